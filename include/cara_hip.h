/* cara_hip.h -- C ABI of libcara_hip.so, the MI355X (gfx950) CaRA hot path.
 *
 * The reference (BonnBytes/CaRA) is pure Python on PyTorch and has no FFI of its own: the
 * functions below are what a maintainer's ctypes binding replaces inside the two patched
 * forwards of /root/reference/src/cara/cara.py (cp_attn :15-60, cp_mlp :63-95) and around the
 * train step of image_classification/vit_cp.py:45-50.  INTEGRATION.md shows that binding.
 *
 * Conventions: plain pointers and sizes only (device pointers unless a comment says host);
 * every entry point enqueues work on `stream` (a hipStream_t passed as void*) and returns an
 * int status: 0 = ok, >0 = CARA_E_* below.  Nothing allocates, frees or synchronises, and nothing
 * keeps state between calls (the one exception is the opt-in diagnostic of cara_profile_sites), so
 * every call is re-entrant and hipGraph-capturable; all work of a call goes to the caller's stream.
 * The caller makes the device of its pointers and stream current (hipSetDevice) before calling.
 * bf16 tensors are raw uint16 (round-to-nearest-even of fp32).  All matrices are row-major and dense
 * unless an ld* is given.
 */
#ifndef CARA_HIP_H
#define CARA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARA_OK 0
#define CARA_E_ARG 1     /* bad shape / null pointer / unsupported size */
#define CARA_E_LAUNCH 2  /* hipGetLastError() != hipSuccess after a launch */

/* ---- library info ------------------------------------------------------------------------ */
int cara_abi_version(void);              /* bumped on any signature change */
const char* cara_build_arch(void);       /* "gfx950" */
/* The 16-bit operand type of this build: "bf16" (libcara_hip.so) or "fp16" (libcara_hip_f16.so: the same sources compiled with
 * IEEE-half operands at the same MFMA rate -- every "bf16" below then reads "fp16"; same ABI, struct for struct).           */
const char* cara_operand_type(void);

/* ---- GEMM with rank-R K-extension -------------------------------------------------------- */
/* C = A[M,K] * B[N,K]^T  (+ A2[M,Rp] * B2[N,Rp]^T)  then an epilogue.  bf16 in, fp32 accumulate.
 * This is the adapter linear of cara.py:25-42,50-58,75-82,87-93 in factored form:
 * A2 = T = X U (cara_skinny_xu), B2 = Vs = s * g (.) V (cara_factor_prep), so that
 * y = X W^T + b + s((X U) (.) g) V^T + s c  costs Rp/K extra MFMA work instead of a second GEMM.
 * Requirements: K % 64 == 0, Rp in {0, 32, 64}; M, N arbitrary (edges are clamped/masked); each
 * operand spans < 4 GiB (32-bit byte offsets), else CARA_E_ARG.                                  */
enum {
  CARA_EPI_BF16 = 0,   /* C bf16 [M,ldc]          = acc + bias                                  */
  CARA_EPI_F32 = 1,    /* C fp32 [M,ldc]          = acc + bias                                  */
  CARA_EPI_GELU = 2,   /* C2 bf16 = u = acc+bias ; C bf16 = gelu_erf(u)           (fc1 forward) */
  CARA_EPI_RESID = 3,  /* C fp32 = aux_f32 + rowscale[m / rows_per_sample] * (acc + bias)       */
  CARA_EPI_DGELU = 4,  /* C bf16 = acc * gelu_erf'(aux_bf16[m,n])                (fc2 backward) */
  /* The same pair with the DERIVATIVE saved instead of the pre-activation: the forward has e = exp(-u^2/2) and the erf polynomial in  */
  /* registers for gelu(u) anyway (4 more VALU per element give gelu'(u) of the UNROUNDED u), and the backward epilogue is one multiply */
  /* instead of ~15 VALU per element of erf arithmetic in a launch whose epilogue is VALU-bound.  gelu' lies in [-0.13, 1.13]: it is kept */
  /* as IEEE half (11 significand bits) in BOTH builds of the library -- bf16's 8 bits would cost more than rounding u did.              */
  CARA_EPI_GELU_DG = 5, /* C bf16 = gelu_erf(u), u = acc+bias ; C2 fp16 = gelu_erf'(u) (NULL: not kept)         (fc1 forward) */
  CARA_EPI_MULH = 6,    /* C bf16 = acc * aux_fp16[m,n]                                                          (fc2 backward) */
};
typedef struct {
  const void* A;  int lda;      /* bf16 [M,K]  */
  const void* B;  int ldb;      /* bf16 [N,K]  */
  const void* A2; const void* B2; int Rp;   /* bf16 [M,Rp], [N,Rp]; Rp = 0 => none */
  int M, N, K;
  const float* bias;            /* fp32 [N] or NULL */
  int epi;
  void* C;  int ldc;
  void* C2;                     /* CARA_EPI_GELU / _GELU_DG only (same ldc); NULL = the pre-activation / derivative is not kept */
  const void* aux;              /* CARA_EPI_RESID: fp32 [M,ldc]; CARA_EPI_DGELU: bf16 [M,ldc]; CARA_EPI_MULH: fp16 [M,ldc] */
  const float* rowscale;        /* CARA_EPI_RESID: fp32 [M / rows_per_sample] or NULL (=1) */
  int rows_per_sample;
  void* scratch;                /* optional: cara_gemm_scratch_bytes() of caller memory, used by ONE call  */
  size_t scratch_bytes;         /* at a time.  With it, few-row products (M <= 128, K >= 512) cut their K  */
                                /* loop into slabs that run as one batched launch (fp32 partials there).   */
  int batch;                    /* > 1: `batch` independent products in ONE launch; product z uses A, B,  */
  long long strideA, strideB, strideC;   /* C advanced by z * stride (elements); no A2/B2/aux/C2 then     */
  /* Whole adapter inside the GEMM (Rp in {32, 64}, B2 set, A2 NULL): every tile also accumulates its rows of */
  /* T = A Ut^T (Ut bf16 [Rp, K], rows >= rank zero: the operand cara_skinny_xu takes), rounds it to bf16 and  */
  /* uses it as the K-extension operand -- the separate skinny pass over A disappears.  The tiles of column   */
  /* 0 also write T [M,Rp] and, if non-NULL, Tt [Rp,ldt] (for cara_tskinny_*).                                */
  const void* Ut;
  void* T_out;
  void* Tt_out;
  int ldt;
  /* Optional second image of B in K-PANEL-MAJOR layout (cara_pack_b_panels): bf16 [K/32][N][32], i.e. the 32     */
  /* columns of K step t of ALL rows are contiguous, so the 128 x 32 operand tile of a K step is one 8-KiB run of  */
  /* full cache lines instead of 128 half lines 2*ldb bytes apart (the L2 -> LDS path delivers ~30 % more tile     */
  /* bytes per second that way, tools/micro/kloop_bw.hip).  Frozen weights are packed once at ingest.  The 32-     */
  /* GEMM reads it INSTEAD of B when non-NULL (B is then only used by the few-row path).                      */
  const void* Bp;
  /* Activations in the same K-panel-major layout (bf16 only; M > 128, no batch).                                    */
  /* a_panels = P > 0: A is [K/32][P][32] (P >= M rows per panel), lda is ignored.      */
  /* c_panels = P > 0 (bf16 epilogues): C is WRITTEN as [N/32][P][32] -- the A operand of the next GEMM, the X of   */
  /* cara_skinny_xu / cara_tskinny_* with ldx = -P -- while C2 / aux keep the row-major ldc.  N % 32 == 0 then.     */
  int a_panels, c_panels;
  /* Optional SECOND B operand, same shape and ldb as B (row-major, bf16): C = A B^T + A B3^T (+ the K-extension), the two  */
  /* products accumulated in fp32 in ONE launch (the K loop runs over B, then over B3).  For a small correction to a large   */
  /* operand -- the exact weight-dropout mode's masked adapter delta against the frozen weight -- that a pre-merged bf16      */
  /* B + B3 would round away.  Plain product only: no Bp, batch, Ut, a_panels or tskinny with it.                            */
  const void* B3;
  /* With Ut: the adapter's rank, if the caller knows it (0 = not stated).  At Rp = 32 and 1 <= Ut_rank <= 16 the kernel computes  */
  /* columns 0 .. 15 of T only (rows >= rank of Ut are zero) and writes columns 16 .. 31 as zeros: same results, bit for bit.      */
  int Ut_rank;
  /* Epilogue riders (CARA_EPI_MULH only; er_Tt == NULL: none).  The tile that has just produced dH = acc * gelu'(u) holds half of what  */
  /* two transposed skinny products of the backward need: dVs = dH^T T of the linear whose dY this C is (fc1), and dU = h^T G' of this    */
  /* linear (fc2) with the h = gelu(u) tile at the same coordinates (er_h: bf16, row-major with C's ldc, or K-panel-major                 */
  /* [N/32][er_h_panels][32]) -- read here once, 16 bytes per lane, instead of 77 MB of dH and 77 MB of h re-read per block by products   */
  /* that ride in the next launch.  Every workgroup multiplies its bf16 dH / h tile, transposed through LDS, by its rows of er_Tt / er_Gt */
  /* (bf16 [>= 16, er_ldg]: T^T and G'^T, rows = the first 16 adapter columns, er_ldg >= M) and writes 16-wide fp32 partial sums for its   */
  /* 128 columns: slab `row tile` of column block n / 64, in the layout of cara_tskinny_partial2_r at rank <= 16 with                     */
  /* cara_gemm_epi_rider_chunks(a) slabs per column block (er_colsum: column sums of dH behind the er_slabs_v slabs, as want_colsum       */
  /* leaves them).  Reduce with cara_ts_reduce::Rc = 16, ::wave_slabs = that chunk count.  Each slab region:                              */
  /* cara_gemm_epi_rider_scratch_bytes(chunks, N).                                                                                         */
  const void* er_Tt;
  const void* er_Gt;
  const void* er_h;
  void* er_slabs_v;
  void* er_slabs_u;
  int er_ldg, er_colsum, er_h_panels;
} cara_gemm_args;
int cara_gemm_bf16(const cara_gemm_args* a, void* stream);
/* Slabs per column block a launch of `a` with epilogue riders writes (= its row tiles), 0: this product cannot carry them (then  */
/* cara_gemm_bf16 / cara_gemm_with_tskinny* return CARA_E_ARG when er_Tt is set).  Needs CARA_EPI_MULH, M > 1024, N >= 3072,      */
/* N % 128 == 0, ldc % 8 == 0, no batch, M % 4 == 0.  Depends on the arguments only.                                              */
int cara_gemm_epi_rider_chunks(const cara_gemm_args* a);
/* The same fields with CARA_EPI_BF16 (er_h NULL, er_Gt / er_slabs_u unused): the dX GEMM of a linear stages that linear's dY as its A   */
/* operand, so its K loop can leave dVs = A^T T (+ the column sums of A with er_colsum) on the way -- from the A tiles in LDS, nothing is */
/* read again.  Slab (row tile of 160, K step of 64) of er_slabs_v, the layout above with K in the place of N; only on the 160 x 256 x 64 */
/* tile: cara_gemm_dv_chunks(a, riders) = the slabs per column block such a launch writes (riders: it also carries a product through      */
/* cara_gemm_with_tskinny_r), 0 = this product cannot (cara_gemm_bf16 then returns CARA_E_ARG).  A K-extension is required (A2 or Ut).    */
int cara_gemm_dv_chunks(const cara_gemm_args* a, int riders);
size_t cara_gemm_epi_rider_scratch_bytes(int chunks, int N);
/* B bf16 [N, K] (row stride ldb) -> out bf16 [K/32][N][32] (cara_gemm_args::Bp); K % 32 == 0 */
int cara_pack_b_panels(const void* B, int ldb, int N, int K, void* out, void* stream);
size_t cara_gemm_scratch_bytes(void);

/* C[z] fp32 [M, ldc] = sum over the rows k of slab z of At[k, m] * Bt[k, n], z = 0 .. nslab - 1, slab z at C + z * slab_stride
 * floats (rows [z * kslab, (z + 1) * kslab) with kslab = K / nslab rounded up to 32).  At bf16 [K, lda >= M], Bt bf16
 * [K, ldb >= N], both ROW-major with the shared index k on the rows: dW = dY^T X (the dense weight gradient of the exact
 * weight-dropout mode) from the activations as they are stored, without transposed copies.  M % 128 == N % 128 == 0,
 * ld % 8 == 0, each operand < 4 GiB; CARA_E_ARG otherwise (callers then transpose and use cara_gemm_bf16).       */
int cara_gemm_tn_f32(const void* At, int lda, const void* Bt, int ldb, float* C, int ldc, int M, int N, int K, int nslab,
                     size_t slab_stride, void* stream);

/* ---- skinny adapter contractions (HBM-bound) --------------------------------------------- */
/* T[M,Rp] = X[M,K] * Ut[Rp,K]^T, bf16 out, also written transposed Tt[Rp,ldt] when Tt != NULL
 * (ldt >= M, multiple of 8).  Forward: T = X U; backward: G' = dY Vs.  K % 32 == 0.
 * In this function and in cara_tskinny_*, a NEGATIVE ldx says that X is K-panel-major: bf16 [K/32][-ldx][32]
 * with -ldx >= M rows per panel (what a GEMM with cara_gemm_args::c_panels = -ldx wrote).                  */
int cara_skinny_xu(const void* X, int ldx, const void* Ut, void* T, void* Tt, int ldt,
                   int M, int K, int Rp, void* stream);
/* The same with the adapter's rank stated (1 <= rank <= Rp; rows >= rank of Ut are zero): at Rp = 32 and rank <= 16 only the
 * first 16 columns are computed (half the Ut fragments and MFMAs of a pass that is bound by what a workgroup pays once) and
 * columns 16 .. 31 of T / rows 16 .. 31 of Tt are written as the zeros they are.  Same results, bit for bit.          */
int cara_skinny_xu_r(const void* X, int ldx, const void* Ut, void* T, void* Tt, int ldt,
                     int M, int K, int Rp, int rank, void* stream);
/* D[K1,Rp] (fp32) = sum_m X[m,K1] * G[m,Rp]  given Gt[Rp,ldg] (= G transposed, bf16); optional
 * colsum[K1] (fp32) = sum_m X[m,:].  Outputs are OVERWRITTEN.  `slabs` is caller scratch of
 * cara_tskinny_scratch_bytes(M, K1, Rp) bytes.  Gives dU = X^T G' and dVs = dY^T T (A.4).      */
size_t cara_tskinny_scratch_bytes(int M, int K1, int Rp);
int cara_tskinny_xtg(const void* X, int ldx, const void* Gt, int ldg, float* D, float* colsum,
                     void* slabs, int M, int K1, int Rp, void* stream);
/* The same in two steps, so that the tiny fixed-order slab sums of many products (the 12 layers
 * of one linear) run as ONE launch: _partial writes the per-block slabs of one product into its
 * own scratch region; _reduce sums `batch` regions spaced slab_stride bytes apart into
 * D[batch,K1,Rp] (and colsum[batch,K1] when non-NULL; every region must then hold column sums). */
int cara_tskinny_partial(const void* X, int ldx, const void* Gt, int ldg, void* slabs, int want_colsum,
                         int M, int K1, int Rp, void* stream);
/* Two products in ONE launch (the dU = X^T G' and dVs = dY^T T of one linear; both Gt share ldg
 * and M; only the second may ask for column sums).                                             */
int cara_tskinny_partial2(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                          const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                          int ldg, int M, int Rp, void* stream);
/* cara_gemm_bf16(a) and cara_tskinny_partial2(...) of the SAME linear as ONE launch: the grid holds the GEMM's tiles
 * and the products' blocks, so the HBM-bound products run under the MFMA-bound GEMM without a second stream (no
 * event between the kernels before and after).  M > 128, no batch; Rp in {32, 64} (64: three workgroups per CU); the
 * products need not be those of the GEMM's own linear (their M may differ from a->M); with a->Ut (the adapter inside
 * the GEMM) only CARA_EPI_BF16 and products of the same Rp.  CARA_E_ARG otherwise (callers then launch the two
 * separately).  The GEMM's epilogue: CARA_EPI_BF16, CARA_EPI_DGELU or CARA_EPI_MULH (the dX products).  Results are bitwise
 * those of the two calls.                                                                                          */
int cara_gemm_with_tskinny(const cara_gemm_args* a, const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                           const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                           int ldg, int M, int Rp, void* stream);
int cara_tskinny_reduce(const void* slabs, size_t slab_stride, float* D, float* colsum, int batch,
                        int M, int K1, int Rp, void* stream);
/* cara_tskinny_partial2 / cara_gemm_with_tskinny with the adapter's rank stated (1 <= rank <= Rp): at Rp = 32 and rank <= 16
 * the products compute the first 16 of their 32 columns only (the others are zero by construction: rows >= rank of Gt are
 * zero) and write 16-WIDE slabs -- reduce them with cara_ts_reduce::Rc = 16 (cara_tskinny_reduce_many), which sums those
 * columns and writes the other columns of D as zeros.  Otherwise identical to the functions without _r.                  */
int cara_tskinny_partial2_r(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                            const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                            int ldg, int M, int Rp, int rank, void* stream);
int cara_gemm_with_tskinny_r(const cara_gemm_args* a, const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                             const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                             int ldg, int M, int Rp, int rank, void* stream);
/* How the products that ride in cara_gemm_with_tskinny_r(a, ..., Rp, rank, ...) leave their partial sums: 0 = one slab per block
 * (every other entry point of this section), 1 = one slab per WAVE, four per block -- the long-K, narrow-N dX products run on a
 * 160 x 256 x 64 tile, one workgroup per CU, whose riding products are streamed by helper waves UNDER the tile's K loop: a helper
 * wave owns one wave's steps of a block and writes its own slab (no combine through LDS).  Pass the answer as
 * cara_ts_reduce::wave_slabs.  cara_tskinny_scratch_bytes() covers both forms.  Depends on the arguments only.            */
int cara_gemm_rider_slab_format(const cara_gemm_args* a, int Rp, int rank);
/* Up to CARA_TS_REDUCE_MAX of those reductions in ONE launch (each entry = the arguments of cara_tskinny_reduce). */
#define CARA_TS_REDUCE_MAX 24
typedef struct {
  const void* slabs; size_t slab_stride; float* D; float* colsum;   /* colsum may be NULL; slabs, D, slab_stride: multiples of 16 bytes */
  int batch, M, K1, Rp;
  int Rc;   /* columns the slabs hold: 0 or Rp = all; 16 (at Rp = 32) = slabs written by the _r functions at rank <= 16 */
  int wave_slabs;   /* 1: the product wrote one slab per WAVE of its blocks (four per block): what cara_gemm_with_tskinny_r does  */
                    /* where cara_gemm_rider_slab_format() says so; 0: one slab per block; >= 2: that many slabs per column      */
                    /* block (the epilogue riders of cara_gemm_args::er_*: cara_gemm_epi_rider_chunks)                            */
} cara_ts_reduce;
int cara_tskinny_reduce_many(const cara_ts_reduce* probs, int n, void* stream);

/* ---- one adapted linear per call (cara_linear_fwd / cara_linear_bwd) -------------------------------- */
/* The adapter linear of cara.py:25-42 (qkv), :50-58 (proj), :75-82 (fc1), :87-93 (fc2) in factored form as ONE call each way:
 * compositions of the entry points above, for an integrator who patches one nn.Linear at a time (cara_vit_forward /
 * cara_vit_backward schedule the same pieces with more fusion).  The per-layer factor images come from cara_factor_prep:
 * Ut [Rp, in] = U^T, U [in, Rp], Vs [out, Rp] = s g (.) V, Vst [Rp, out] (bf16, columns / rows >= rank zero), bias = b + s c.  */
typedef struct {
  const void* W;  const void* Wt;     /* bf16 [out, in] frozen weight; [in, out] its transpose (backward only)             */
  const void* Ut; const void* U;      /* bf16 [Rp, in], [in, Rp]                                                           */
  const void* Vs; const void* Vst;    /* bf16 [out, Rp], [Rp, out]                                                         */
  const float* bias;                  /* fp32 [out] or NULL                                                                */
  int in, out, Rp, rank;              /* Rp in {32, 64}, 1 <= rank <= Rp                                                   */
} cara_linear;
/* Y = epilogue(X W^T + bias + (X U) Vs^T): T [M, Rp] = X U (bf16) and its transpose Tt [Rp, ldt] (ldt >= M rounded up to 32,
 * multiple of 8) are WRITTEN -- the backward reads Tt.  epi / Y / Y2 / aux / rowscale / rows_per_sample: the epilogue fields of
 * cara_gemm_args -- C, C2, aux, ...; ldy = 0: dense.  X bf16 [M, ldx].                                                     */
int cara_linear_fwd(const cara_linear* L, const void* X, int ldx, int M, void* T, void* Tt, int ldt, int epi, void* Y, int ldy,
                    void* Y2, const void* aux, const float* rowscale, int rows_per_sample, void* stream);
/* Given dY bf16 [M, lddy], the saved input X and the forward's Tt: G [M, Rp] = dY Vs and Gt (scratch, written);
 * dX bf16 [M, lddx] = dY W + G U^T (skipped when NULL); dU fp32 [in, Rp] = X^T G, dVs fp32 [out, Rp] = dY^T T, dc fp32 [out] =
 * column sums of dY (NULL: not wanted) -- the gradients cara_factor_grad_reduce takes per layer.  slabs_u / slabs_v:
 * cara_tskinny_scratch_bytes(M, in, Rp) / (M, out, Rp) bytes of caller scratch; they, dU and dVs: 16-byte aligned.         */
int cara_linear_bwd(const cara_linear* L, const void* dY, int lddy, const void* X, int ldx, const void* Tt, int M, void* G, void* Gt,
                    int ldt, void* dX, int lddx, void* slabs_u, void* slabs_v, float* dU, float* dVs, float* dc, void* stream);

/* ---- LayerNorm (eps inside the sqrt, biased variance; timm Block norm1/norm2/norm) ------- */
/* y bf16 [M,C] = (x - mean) * rstd * gamma + beta; saves mean, rstd fp32 [M].  x fp32 rows with
 * stride ldx elements (ldx = tokens*C picks the cls rows for the final norm).  C % 256 == 0.   */
int cara_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int M, int C, float eps, void* stream);
/* dx_out fp32 = dx_in + LN'(dy) ; optional dyb bf16 = rowscale[m / rows_per_sample] * dx_out.
 * dy is bf16 [M,C]; x/dx rows strided by ldx like the forward.  dx_in may be NULL (= 0) and may
 * alias dx_out.                                                                               */
int cara_layernorm_bwd(const void* dy, const float* x, long ldx, const float* gamma,
                       const float* mean, const float* rstd, const float* dx_in, float* dx_out,
                       void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                       void* stream);
/* The same with the skinny adapter product of the NEXT linear fused in (its input row is in registers
 * here, so the separate cara_skinny_xu pass and its re-read of the activations disappear):
 *   _fwd_xu: T[M,Rp] = y Ut^T (Ut bf16 [Rp,C], rows >= rank zero), Tt[Rp,ldt] its transpose (or NULL);
 *   _bwd_xu: G[M,Rp] = dyb Vst^T with dyb the row-scaled bf16 gradient this kernel emits (dyb may be NULL
 *            when only G is wanted), Gt its transpose.  Same rounding points as cara_skinny_xu.
 * Rp in {32, 64} (rank <= Rp), C in {256, 768, 1024}; else CARA_E_ARG (callers then keep the separate
 * cara_skinny_xu pass).                                                                                  */
int cara_layernorm_fwd_xu(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                          float* mean, float* rstd, int M, int C, float eps, const void* Ut, int rank,
                          int Rp, void* T, void* Tt, int ldt, void* stream);
int cara_layernorm_bwd_xu(const void* dy, const float* x, long ldx, const float* gamma,
                          const float* mean, const float* rstd, const float* dx_in, float* dx_out,
                          void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                          const void* Vst, int rank, int Rp, void* G, void* Gt, int ldt, void* stream);
/* The general forms: Ut / Vst may be NULL (no fused product), and the bf16 output (y, resp. dyb) is written
 * K-panel-major -- [C/32][panels][32] with panels >= M rows per panel, the layout cara_gemm_args::a_panels
 * and a negative ldx of the skinny products read -- when y_panels / dyb_panels > 0 (0 = row-major).        */
int cara_layernorm_fwd_ex(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                          float* mean, float* rstd, int M, int C, float eps, const void* Ut, int rank,
                          int Rp, void* T, void* Tt, int ldt, int y_panels, void* stream);
int cara_layernorm_bwd_ex(const void* dy, const float* x, long ldx, const float* gamma,
                          const float* mean, const float* rstd, const float* dx_in, float* dx_out,
                          void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                          const void* Vst, int rank, int Rp, void* G, void* Gt, int ldt, int dyb_panels,
                          void* stream);

/* ---- attention (cara.py:43-48; softmax(q k^T * scale) v per head) ------------------------- */
/* qkv bf16 [B*N, 3*H*64] with column k*H*64 + h*64 + d (k = q,k,v): exactly the layout
 * cara.py:39 reshapes.  out bf16 [B*N, H*64]; lse fp32 [B,H,N].  N <= 608, head dim 64 (N <= 224:   *
 * score rows in registers; above: two sweeps over the key tiles, whole K/V of a head in LDS).  */
int cara_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale,
                       void* stream);
int cara_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                       void* dqkv, int B, int N, int H, float scale, void* stream);
/* The same for the cls query alone (row 0 of every sample) -- all the LAST block needs: only the cls row of its attention
 * output reaches the logits (timm's `x[:, 0]` behind the final norm; cara.py:43-48 computes every row).  _fwd writes
 * out[b N + 0, :] and lse[b, h, 0] and leaves the other rows alone; _bwd reads those rows (and dout[b N + 0, :]) and writes
 * ALL of dqkv: dK / dV of every key row, dQ of the cls row, zeros for the dQ of the other rows.  Same rounding points as the
 * full kernels.  N <= 608.                                                                                             */
int cara_attention_cls_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream);
int cara_attention_cls_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                           void* dqkv, int B, int N, int H, float scale, void* stream);

/* ---- small pieces of the ViT around the blocks -------------------------------------------- */
/* patches bf16 [B*gh*gw, C*p*p] <- images fp32 [B,C,Hi,Wi]  (Conv2d k=s=p as a GEMM operand)  */
int cara_im2col_patches(const float* img, void* patches, int B, int C, int Hi, int Wi, int p,
                        void* stream);
/* x fp32 [B,1+P,D]: row 0 = cls + pos[0]; row 1+i = emb[b*P+i] + pos[1+i]                      */
int cara_assemble_tokens(const float* emb, const float* cls, const float* pos, float* x, int B,
                         int P, int D, void* stream);
/* loss[0] = mean cross-entropy over B (loss must hold 1 + B floats: [1..B] is scratch);
 * dlogits (optional) = (softmax - onehot)/B ; logits fp32 [B,C]   (vit_cp.py:47)              */
int cara_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                       int B, int C, void* stream);
/* The same with the gradient pre-scaled, so that a train step needs no separate scaling launches:
 *   dlogits = (softmax - onehot)/B * dscale * (loss_scale ? *loss_scale : 1)
 * dscale (host) = 1/world_size of a data-parallel job (the all-reduce is then a plain SUM); loss_scale (device float,
 * may be NULL) = the dynamic loss scale of the IEEE-half operand build (cara_amp_update).  found_inf (device float, may
 * be NULL) is set to 0 here: the backward's final gradient writes raise it (cara_vit_shape::found_inf).  loss[0] is the
 * UNSCALED mean loss.                                                                                                  */
int cara_cross_entropy_ex(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                          int B, int C, float dscale, const float* loss_scale, float* found_inf, void* stream);
/* bf16 <-> fp32 helpers (weight ingest)                                                        */
int cara_f32_to_bf16(const float* src, void* dst, size_t n, void* stream);
int cara_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream);
/* the same for activation-sized matrices: row strides lds >= cols, ldd >= rows (multiples of 8), 16-byte accesses */
int cara_transpose_bf16_ld(const void* src, long lds, void* dst, long ldd, int rows, int cols, void* stream);

/* ---- CP factor preparation / gradient scatter --------------------- */
/* Geometry of one adapted ViT.  The reference hard-codes dim 768 / heads 12 / depth 12
 * (cara.py:112-125); the same formulas are kept general.                                      */
typedef struct {
  int depth, dim, heads, rank, Rp;
  float scale;                       /* child.s, cara.py:149,159 */
  /* Order of the CP tensorisation of the QKV adapter (image_classification/dim_experiment.py:188-207,264-295).
   * 0 or 4 (src/cara, the default): dW[k,e,h,d] = sum_r R1 A1[3l+k] A2[e] A3[h] A4[d],  A1 [3 depth,R], A2 [dim,R],
   *         A3 [heads,R], A4 [dim/heads,R].
   * 3: dW[k,e,o] = sum_r R1 A1[3l+k] A2[e] A3[o],  A3 [dim,R], no A4.
   * 5: dW[k,e,h,d] = sum_r R1 A1[l] A2[k] A3[e] A4[h] A5[d],  A1 [depth,R], A2 [3,R], A3 [dim,R], A4 [heads,R],
   *    A5 [dim/heads,R].
   * All three are rank-R in (in, out) and run on the same factored kernels; only the factor pack and the
   * gradient scatter differ.
   * 2: dW[k, e * dim + o] = sum_r R1 A1[3l+k] A2[e * dim + o],  A2 [dim * dim, R], no A3 / A4 (NULL): a sum of R DENSE
   *    dim x dim matrices per projection, not low-rank -- the QKV linear then runs in the dense-delta form
   *    (cara_dense_delta_* below; weight_dropout off only) while proj / fc1 / fc2 keep the factored kernels.           */
  int cp_length;
} cara_geom;
/* Pointers to the CP tensors (fp32, shapes of cara.py:112-125 for cp_length 4) or to their gradients.
 * A4 is unused for cp_length 3, A5 only used for cp_length 5 (NULL otherwise).                               */
typedef struct {
  float *A1, *A2, *A3, *A4, *P1, *P2, *P3, *R1, *R2, *bias1, *bias2, *bias3;
  float *A5;
} cara_cp;
/* Per-layer operand pack written by cara_factor_prep (bf16 unless noted), l = 0..depth-1:
 *   Ut_* [Rp,in] (U transposed, K-contiguous), U_* [in,Rp], Vs_* [out,Rp], Vst_* [Rp,out]
 * laid out back to back; offsets via cara_pack_offsets().  Row padding rank..Rp-1 is zero.     */
typedef struct {
  size_t Ut_qkv, U_qkv, Vs_qkv, Vst_qkv;
  size_t Ut_proj, U_proj, Vs_proj, Vst_proj;
  size_t Ut_fc1, U_fc1, Vs_fc1, Vst_fc1;
  size_t Ut_fc2, U_fc2, Vs_fc2, Vst_fc2;
  size_t bias_proj, bias_fc1, bias_fc2;      /* fp32: b + s*CP_bias (needs backbone bias)      */
  size_t layer_stride;                       /* bytes between consecutive layers               */
  size_t total;                              /* bytes for all layers                           */
} cara_pack_layout;
int cara_pack_offsets(const cara_geom* g, cara_pack_layout* out);            /* host only */
/* Build every layer's operand pack from the CP tensors in one launch.  base_bias_* are the
 * frozen fp32 biases of proj/fc1/fc2, [depth, out] contiguous.                                 */
int cara_factor_prep(const cara_geom* g, const cara_cp* cp, const float* base_bias_proj,
                     const float* base_bias_fc1, const float* base_bias_fc2, void* pack,
                     void* stream);
/* Scatter per-layer skinny gradients onto the 12 shared tensors (A.4).  dU_x / dVs_x are fp32
 * [depth, in|out, Rp]; dc_x fp32 [depth, out] (column sums of dY of proj/fc1/fc2).  grads are
 * OVERWRITTEN.  scratch: cara_factor_grad_scratch_bytes(g) bytes (partials, summed in a fixed
 * order: bitwise reproducible).                                                                */
typedef struct {
  const float *dU_qkv, *dVs_qkv, *dU_proj, *dVs_proj, *dU_fc1, *dVs_fc1, *dU_fc2, *dVs_fc2;
  const float *dc_proj, *dc_fc1, *dc_fc2;
} cara_layer_grads;
size_t cara_factor_grad_scratch_bytes(const cara_geom* g);
int cara_factor_grad_reduce(const cara_geom* g, const cara_cp* cp, const cara_layer_grads* lg,
                            const cara_cp* grads, void* scratch, void* stream);
/* The same under a loss scale: every gradient written is multiplied by 1 / *loss_scale (device float; NULL = 1), and
 * *found_inf (device float; NULL = no check) is set to 1 when a written value is not finite.                          */
int cara_factor_grad_reduce_ex(const cara_geom* g, const cara_cp* cp, const cara_layer_grads* lg,
                               const cara_cp* grads, void* scratch, const float* loss_scale, float* found_inf,
                               void* stream);

/* ---- order-2 QKV tensorisation (image_classification/dim_experiment.py:203-207,293-297; cara_geom::cp_length == 2) ---- */
/* Dm bf16 [depth][3 dim, dim] with Dm[l][k dim + o][e] = s * sum_r R1[r] A1[3l+k, r] A2[e dim + o, r] (the second B operand,
 * cara_gemm_args::B3, of the QKV linear next to the frozen weight) and Dmt bf16 [depth][dim, 3 dim] its transpose (for dX). */
int cara_dense_delta_materialize(const cara_geom* g, const cara_cp* cp, void* Dm, void* Dmt, void* stream);
/* From dD fp32 [depth][3][dim (e), dim (o)] = x^T dY_k per block and projection (cara_gemm_tn_f32 + cara_sum_slabs_f32) to the
 * gradients of CP_A1, CP_A2, CP_R1 (OVERWRITTEN in `grads`; run it AFTER cara_factor_grad_reduce, which for this order leaves
 * CP_A1 / CP_A2 alone and writes a zero CP_R1).  Fixed summation order.  scratch: cara_dense_delta_grad_scratch_bytes(g). */
size_t cara_dense_delta_grad_scratch_bytes(const cara_geom* g);
int cara_dense_delta_grad(const cara_geom* g, const cara_cp* cp, const float* dD, const cara_cp* grads, void* scratch, void* stream);
/* out[j] = sum_z slabs[z * slab_stride + j], j < count (fixed order): the split-K slabs of cara_gemm_tn_f32 into one matrix */
int cara_sum_slabs_f32(const float* slabs, int nslab, size_t slab_stride, size_t count, float* out, void* stream);

/* ---- optimiser step (image_classification/vit_cp.py:185 `torch.optim.AdamW(trainable, ...)`, :50 `opt.step()`) ---- */
/* torch.optim.AdamW's arithmetic (amsgrad off, maximize off) over up to 32 fp32 tensors in ONE launch:
 *   p *= 1 - lr wd;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;
 *   p -= lr / bias_correction1 * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with bias_correction1 = 1 - beta1^step and bias_correction2_sqrt = sqrt(1 - beta2^step) computed by the caller (in double, as
 * torch does on the host).  `group` picks the tensor's lr / weight_decay (param groups; betas and eps are shared).
 * The struct travels by value into the kernel: nothing lives in device memory, a changed lr costs nothing.          */
#define CARA_ADAMW_MAX_TENSORS 32
#define CARA_ADAMW_MAX_GROUPS 4
typedef struct {
  float* p; const float* g; float* m; float* v;   /* parameter, gradient, exp_avg, exp_avg_sq: n fp32 each */
  size_t n;
  int group;
} cara_adamw_tensor;
typedef struct {
  cara_adamw_tensor t[CARA_ADAMW_MAX_TENSORS];
  int ntensors;
  int step;                                        /* 1-based step count (validation only; the corrections come below) */
  float lr[CARA_ADAMW_MAX_GROUPS], weight_decay[CARA_ADAMW_MAX_GROUPS];
  float one_minus_beta1, beta2, one_minus_beta2, eps;
  float bias_correction1, bias_correction2_sqrt;
  /* device float or NULL: when *skip_flag != 0 the launch changes nothing (a step whose gradients overflowed under the
   * loss scale of the IEEE-half build: cara_vit_shape::found_inf, all-reduced with the gradients)                     */
  const float* skip_flag;
  /* device float[1 + CARA_ADAMW_MAX_GROUPS] or NULL: { step count t, lr of group 0, 1, ... }.  When given, the kernel takes the
   * learning rates from it and computes both bias corrections from t itself (`step`, `lr`, `bias_correction*` above are ignored):
   * nothing that changes from step to step travels in the launch arguments, so a launch captured into a hipGraph stays correct
   * under replay -- the host rewrites the five floats (stream-ordered) before each replay.                                     */
  const float* dyn;
} cara_adamw_args;
int cara_adamw_step(const cara_adamw_args* a, void* stream);
/* Dynamic loss scale of the IEEE-half operand build, entirely on the device (no host synchronisation anywhere in a step).
 * state (device float[4]) = { scale, clean steps since the last change, steps skipped so far, unused }.  Once per step,
 * AFTER the all-reduce and BEFORE the optimiser: *found_inf != 0 (some rank wrote a non-finite gradient) -> scale *= backoff,
 * counter = 0, skipped += 1; otherwise counter += 1 and, at `interval` clean steps, scale *= growth (capped at max_scale),
 * counter = 0.  torch.amp.GradScaler's rule.  found_inf is left as it is (cara_adamw_args::skip_flag reads it next).    */
int cara_amp_update(float* state, const float* found_inf, float growth, float backoff, int interval, float max_scale,
                    void* stream);

/* ---- exact weight-space dropout mode (the reference's train-mode arithmetic, cara.py:35,57,81,92) ---- */
/* keep(o,i) of linear `linear_id` = (cara_weight_dropout_hash(o*in + i, seed, linear_id) >> 8) >= p * 2^24
 * (host-callable mirror of the device hash, so that tests and the oracle can rebuild the masks).             */
unsigned cara_weight_dropout_hash(unsigned idx, unsigned seed, unsigned linear_id);
/* Weff bf16 [out,in] = W + keep/(1-p) * (Vs U^T): W bf16 [out,in], U bf16 [in,Rp], Vs bf16 [out,Rp] (= s g (.) V,
 * the operand pack of cara_factor_prep).  p = 0 is the eval-time merge of the reference's dW into W.
 * W == NULL: only the masked delta keep/(1-p) * (Vs U^T) is written -- the B3 operand of cara_gemm_args, which keeps
 * an adapter far below W's bf16 ulp (zero-initialised factors early in training) from being rounded away.        */
int cara_materialize_merge(const void* W, const void* U, const void* Vs, int Rp, int out, int in, float p,
                           unsigned seed, unsigned linear_id, void* Weff, void* stream);
/* From the dense weight gradient dW (= dY^T X, given as `nslab` split-K partial slabs of fp32 [out,in], slab_stride
 * floats apart, summed on the fly) to the skinny quantities cara_factor_grad_reduce takes:
 * dVs[o,r] = sum_i keep/(1-p) dW[o,i] U[i,r], dU[i,r] = sum_o keep/(1-p) dW[o,i] Vs[o,r] (fp32 [out,Rp] / [in,Rp],
 * overwritten, fixed summation order; ONE pass over dW for both).  scratch: cara_dropout_grad_scratch_bytes(out, in, Rp). */
size_t cara_dropout_grad_scratch_bytes(int out, int in, int Rp);
int cara_dropout_grad_contract(const float* dW, int nslab, size_t slab_stride, const void* U, const void* Vs, int Rp,
                               int out, int in, float p, unsigned seed, unsigned linear_id, float* dU, float* dVs,
                               void* scratch, void* stream);
/* out fp32 [N] = column sums of a bf16 [M, ld] matrix (dc = sum_m dY), fixed order; scratch:
 * cara_colsum_scratch_bytes(N).                                                                               */
size_t cara_colsum_scratch_bytes(int N);
int cara_colsum_bf16(const void* X, int ld, int M, int N, float* out, void* scratch, void* stream);

/* ---- whole adapted ViT: forward and backward as stream-ordered kernel sequences ------------ */
/* What model(x) / loss.backward() of vit_cp.py:46-49 run, for the factored adapters.  Both calls
 * only enqueue kernels on `stream` (no allocation, no sync): capturable in a hipGraph.         */
typedef struct {           /* frozen backbone (timm 0.4.12 VisionTransformer), device pointers  */
  const void* patch_w;     /* bf16 [dim, chans*patch*patch]   (patch_embed.proj.weight)         */
  const float* patch_b;    /* fp32 [dim]                                                        */
  const float* cls;        /* fp32 [dim]                      (cls_token)                       */
  const float* pos;        /* fp32 [tokens, dim]              (pos_embed)                       */
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;   /* fp32 [depth, dim]                            */
  const void *qkv_w, *qkv_wt;   const float* qkv_b;    /* bf16 [depth,3dim,dim], [depth,dim,3dim] */
  const void *proj_w, *proj_wt; const float* proj_b;   /* bf16 [depth,dim,dim] both              */
  const void *fc1_w, *fc1_wt;   const float* fc1_b;    /* bf16 [depth,4dim,dim], [depth,dim,4dim] */
  const void *fc2_w, *fc2_wt;   const float* fc2_b;    /* bf16 [depth,dim,4dim], [depth,4dim,dim] */
  const float *norm_g, *norm_b; /* fp32 [dim]                                                    */
  /* optional (NULL = unused): the eight matrices above once more as K-panel-major images, per layer
   * [K/32][N][32] with N x K the shape of the row-major matrix (cara_pack_b_panels, cara_gemm_args::Bp) */
  const void *qkv_wp, *qkv_wtp, *proj_wp, *proj_wtp, *fc1_wp, *fc1_wtp, *fc2_wp, *fc2_wtp;
} cara_vit_weights;
typedef struct {
  int B, img, patch, chans, tokens, num_classes;
  float eps;
  /* weight-space dropout on the materialised adapters (cara.py:35,57,81,92).  0 = off: the factored path
   * (eval semantics, and the fast training default: the mask does not factor).  1 = exact: every
   * linear runs on W_eff = W + keep/(1-p) dW (cara_materialize_merge, linear id 4*layer + {qkv,proj,fc1,fc2},
   * seed `wd_seed`, which the caller changes every step) and the adapter gradients come from the dense
   * dW = dY^T X.  The workspace is larger in this mode (cara_vit_workspace_bytes sees the flag).           */
  int wd_exact;
  float wd_p;
  unsigned wd_seed;
  /* 1: no backward will follow this forward (eval / no_grad).  The forward then skips what only the backward
   * reads -- the bf16 pre-activation of fc1 (77 MB per block at bs 64), T^T of the adapter products -- and
   * cara_vit_backward on that workspace is an error until a forward with inference = 0 has run.            */
  int inference;
  /* Loss scaling (the IEEE-half operand build; both NULL otherwise).  cara_vit_backward is linear in dlogits: a caller that
   * scaled dlogits by S = *loss_scale (device float) gets every gradient it writes multiplied by 1/S here, in the kernels
   * that write them (no separate unscale launch), and *found_inf (device float, zeroed by the caller or by
   * cara_cross_entropy_ex) is set to 1 when a written gradient is not finite.                                           */
  const float* loss_scale;
  float* found_inf;
} cara_vit_shape;
size_t cara_vit_workspace_bytes(const cara_geom* g, const cara_vit_shape* s);
/* Both calls are stateless (everything lives in the caller's workspace and runs in order on the caller's stream):
 * any number of workspaces / streams / host threads may be in flight at once, one workspace per concurrent call.
 * images fp32 [B,chans,img,img]; droppath fp32 [depth,2,B] per-sample branch multipliers
 * (mask/keep_prob of timm DropPath) or NULL; head_w fp32 [classes,dim], head_b fp32 [classes];
 * logits fp32 [B,classes].  The workspace must be zero-filled once before its first use and
 * must not be touched between a forward and its backward.                                      */
int cara_vit_forward(const cara_geom* g, const cara_vit_shape* s, const cara_vit_weights* w,
                     const cara_cp* cp, const float* head_w, const float* head_b, const float* images,
                     const float* droppath, void* workspace, float* logits, void* stream);
/* dlogits fp32 [B,classes] -> grads of the 12 CP tensors (overwritten), dhead_w, dhead_b.       */
int cara_vit_backward(const cara_geom* g, const cara_vit_shape* s, const cara_vit_weights* w,
                      const cara_cp* cp, const float* head_w, const float* dlogits, const float* droppath,
                      void* workspace, const cara_cp* grads, float* dhead_w, float* dhead_b, void* stream);
/* classifier head backward on its own (used by the two above):
 * dW[c,d] = sum_b dl[b,c] xn[b,d]; db[c] = sum_b dl[b,c]; dxn bf16 [B,D] = dl W               */
int cara_head_backward(const float* dlogits, const void* xn_bf16, const float* head_w, float* dhead_w,
                       float* dhead_b, void* dxn_bf16, int B, int classes, int D, void* stream);
/* final LayerNorm of the B cls rows (row stride ldx floats) + classifier head, in fp32 throughout:
 * logits fp32 [B,classes] = LN(x) head_w^T + head_b; also writes LN(x) as 16-bit xn [B,D] and mean / rstd [B]
 * (what cara_head_backward and the final LayerNorm's backward read).  The head is 0.1 % of the work; rounding its two
 * operands to 16 bits was 2.8e-4 (fp16) / 2.4e-3 (bf16) of logit error for nothing (tools/fp16_sim_study.py).        */
int cara_head_forward(const float* x, long ldx, const float* gamma, const float* beta, const float* head_w,
                      const float* head_b, void* xn_16, float* mean, float* rstd, float* logits, int B, int classes,
                      int D, float eps, void* stream);

/* ---- ABI self-description ------------------------------------------------------------------- */
/* sizeof() of the structs above as THIS library was compiled, so that a binding can assert that its own mirror of a
 * struct (ctypes.Structure, cgo, JNA ...) has the same size before it hands one over: a field missing at the end of
 * the mirror would otherwise make the library read past it.  cara_sizeof_struct(CARA_STRUCT_*) returns 0 for an
 * unknown id.                                                                                                   */
enum {
  CARA_STRUCT_GEMM_ARGS = 0, CARA_STRUCT_GEOM, CARA_STRUCT_CP, CARA_STRUCT_PACK_LAYOUT, CARA_STRUCT_LAYER_GRADS,
  CARA_STRUCT_VIT_WEIGHTS, CARA_STRUCT_VIT_SHAPE, CARA_STRUCT_TS_REDUCE, CARA_STRUCT_LINEAR, CARA_STRUCT_ADAMW_ARGS,
  CARA_STRUCT_COUNT
};
size_t cara_sizeof_struct(int which);
size_t cara_sizeof_gemm_args(void);      /* == cara_sizeof_struct(CARA_STRUCT_GEMM_ARGS) */

/* ---- the step's one collective (optional) ------------------------------------------------------ */
/* SUM all-reduce, in place, of `count` fp32 values on `stream` through RCCL: the flat gradient buffer of a data-parallel
 * step (the reference: DistributedDataParallel's bucketed all-reduce, image_classification/vit_cp.py's launcher; here ONE
 * call per step -- the buffer is 121 923 gradients + the found-inf word at the headline configuration, pre-scaled by
 * 1 / world size at the cross-entropy, cara_cross_entropy_ex).  nccl_comm = the caller's ncclComm_t.  librccl is looked up at
 * the first call: CARA_E_LAUNCH when the process has none.  The Python side of this repository uses torch.distributed
 * (backend "nccl" = RCCL), whose communicator is not exposed; a C++ host that owns one calls this.                      */
int cara_allreduce_flat(void* nccl_comm, float* buf, size_t count, void* stream);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* HIP-event brackets around the kernels of chosen call sites INSIDE cara_vit_forward / cara_vit_backward, recorded
 * on the compute stream, so that a benchmark can read per-kernel launch durations from within its timed region
 * (bench.py: `roofline`).  mask = bit set of CARA_SITE_* (0 = off); only the full-size blocks l with l % every == every / 2
 * are bracketed (every = 1: all of them; never block 0 otherwise -- its qkv dX and LayerNorm-1 backward do not exist).  Every bracket idles the chip for ~15 us (three event records), so a timed run brackets one site
 * on a few layers per step.  Read after synchronising: avg_ms = mean duration of the last (up to CARA_SITE_RING)
 * brackets of that site minus marker_ms, the mean of the EMPTY bracket recorded behind each (the markers' own
 * cost); launches = brackets recorded since cara_profile_sites() (0: the site never ran).  A GEMM site holds exactly the GEMM launch (with the
 * transposed skinny products that ride in it), a SKINNY site the separate cara_skinny_xu passes.
 * This is the ONLY process-global state of the library: diagnostic, off by default, not thread-safe.            */
enum {
  CARA_SITE_QKV_FWD = 0, CARA_SITE_PROJ_FWD, CARA_SITE_FC1_FWD, CARA_SITE_FC2_FWD,
  CARA_SITE_QKV_BWD, CARA_SITE_PROJ_BWD, CARA_SITE_FC1_BWD, CARA_SITE_FC2_BWD,
  CARA_SITE_ATTN_FWD, CARA_SITE_ATTN_BWD,
  CARA_SITE_LN1_FWD, CARA_SITE_LN2_FWD, CARA_SITE_LN1_BWD, CARA_SITE_LN2_BWD,
  CARA_SITE_SKINNY_FWD, CARA_SITE_SKINNY_BWD,
  CARA_SITE_COUNT
};
#define CARA_SITE_RING 64
int cara_profile_sites(unsigned long long mask, int every);
int cara_profile_site_read(int site, float* avg_ms, float* marker_ms, int* launches);   /* host pointers */
/* One ds_read_b64_tr_b16 per lane (64 lanes) over an LDS image of 16-bit values sm[i] = i, lane l
 * reading at byte address byte_addr[l] (device int[64], multiples of 8, < 16384); out = device
 * short[256] (4 per lane).  Pins the lane semantics the attention kernels rely on.             */
int cara_debug_tr_probe(const int* byte_addr, short* out, void* stream);
/* The attention kernels' transposed fragment of a staged [N,64] bf16 matrix: out bf16 [64 lanes][8]. */
int cara_debug_tr_frag(const void* src, void* out, int N, int cbase, int base, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARA_HIP_H */
