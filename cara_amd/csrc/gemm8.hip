// bf16 MFMA GEMM, MT x 256 x 64 tile, ONE 512-thread workgroup per CU (gfx950 / MI355X only).
//
//   C = A[M,K] . B[N,K]^T  (+ A2[M,32] . B2[N,32]^T)  -> the epilogues of gemm_epilogue.h
//
// The second tile family of the library, for the products whose K loop is long against their output: the adapted
// linears with N = dim (/root/reference/src/cara/cara.py:50 proj, :87 fc2 forward; the dX of :75 fc1 and :25 qkv),
// i.e. K = 3072 / 2304 at N = 768.  The 128 x 128 x 32 kernel of gemm.hip stages 16 KiB per 1.05 MFLOP and runs
// at what a CU's vector-memory path delivers; this tile stages (MT + 256) x 128 B per K step of 64
// -- 0.65 x the bytes per flop at MT = 160 -- and keeps the staging in flight ACROSS its barriers:
//
//  * 8 waves as 2 (M) x 4 (N): a wave owns (MT / 2) x 64 outputs = RT x 4 accumulators of v_mfma_f32_16x16x32_bf16
//    (RT = MT / 32: 5 at MT = 160, 8 at MT = 256).  Waves w and w + 4 share a SIMD and run ONE BARRIER APART: while one
//    issues a cluster of MFMAs the other issues its LDS fragment reads and its share of the LDS-DMA, so every SIMD's
//    matrix pipe has a wave feeding it in every barrier interval.
//  * a K step of 64 is FOUR phases; a phase = {fragment reads of one register sub-tile, one sub-buffer of LDS-DMA for
//    a later K step, counted s_waitcnt vmcnt (never 0 in the steady state), s_barrier, lgkmcnt(0), MFMAs of one
//    quadrant of the wave's outputs under s_setprio, s_barrier}.  Quadrants: (A rows sub0 | sub1) x (B columns sub0 | sub1).
//  * LDS: two K-step buffers, each four sub-buffers A0 | A1 | B0 | B1 (the rows of the sub-tile of BOTH wave rows /
//    all four wave columns), rows of 128 B = whole cache lines of the row-major operands, the 16-byte chunk index
//    XOR-swizzled with (row & 7) on the global SOURCE address (the DMA destination is lane-linear) and again on the
//    ds_read_b128 address: conflict-free.  A sub-buffer is read in exactly one phase of its K step and restaged two or
//    three phases later (the stagger of the wave rows is why not one), five phases = 1.25 K steps ahead of its reads;
//    four sub-buffers (a whole K step, 52 KiB at MT = 160) are in flight behind every wait.
//  * the K-extension ([T | Vs], Rp = 32: the CaRA adapter term) is two more sub-buffers of 64-byte rows
//    issued in the DMA slots the last K steps leave free, and one more cluster of MFMAs.
//  * the WHOLE adapter inside (cara_gemm_args::Ut, rank <= 16): a 2-KiB slab of Ut per K step (one more DMA instruction per wave and
//    K step: waves 0 and 1 move the slab, the others an instruction that moves nothing, so that every wave counts the same), and
//    every wave accumulates TWO 16 x 16 tiles of T = A Ut^T -- one row tile of each sub-tile, re-read from LDS at a wave-dependent
//    address: +4 MFMAs on 40 per K step --; T is rounded to bf16 into the extension's LDS image, the tiles of column 0 write T / Tt.
//  * the transposed skinny products a dX launch carries (cara_gemm_with_tskinny_r) are 512-thread workgroups of two tskinny blocks
//    behind the tiles: they take the CUs the tiles leave free and then the CUs whose tile is done.
//  * M = 12608 rows: 79 tiles of 160 rows x 3 column tiles = 237 workgroups = 92.6 % of the CUs in ONE round.
//
// Sequence numbers of the sub-buffers (what the counted waits are derived from): K step t holds seq 4t .. 4t + 3 =
// A0, B0, B1, A1; phase phi = 4t + (p - 1) issues seq phi + 6 and then waits until seq <= phi + 2 has landed (read
// one phase later), i.e. allows seq phi + 3 .. phi + 6 to stay in flight -- always one sub-buffer of each kind.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "gemm8.h"
#include "tskinny_body.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int N>
__device__ __forceinline__ void g8_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g8_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void g8_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int RT0_, int RT1_>
struct G8 {
  static constexpr int RT0 = RT0_, RT1 = RT1_, RT = RT0_ + RT1_;
  static constexpr int MT = 32 * RT;                       // rows per tile (two wave rows of 16 RT)
  static constexpr int A0_ROWS = 32 * RT0, A1_ROWS = 32 * RT1;
  static constexpr int A0_OFF = 0, A1_OFF = A0_ROWS * 128, B0_OFF = MT * 128, B1_OFF = B0_OFF + 128 * 128;
  static constexpr int BUF = MT * 128 + 2 * 128 * 128;     // one K step: (MT + 256) rows of 128 B
  static constexpr int U_OFF = 2 * BUF;                    // adapter inside: two 2-KiB slabs of Ut (16 rows x 128 B), one per buffer
  static constexpr int JUNK_OFF = U_OFF + 2 * 2048;        // 1 KiB: where the DMA instructions that move nothing point
  static constexpr int LDS = JUNK_OFF + 1024;
  static constexpr int NPA0 = A0_ROWS / 8, NPA1 = A1_ROWS / 8;   // one-KiB pieces (8 rows x 128 B)
  static_assert(NPA0 % 8 == 0 || NPA0 % 8 == 4, "piece counts: two wave classes at most");
  static_assert(NPA1 % 8 == 0 || NPA1 % 8 == 4, "piece counts: two wave classes at most");
  static_assert(A0_ROWS * 128 >= MT * 64, "the extension's T rows (64 B each) go to the A0 region");
};
// LDS-DMA instructions per wave and sub-buffer: waves 0..3 (CLS 0) issue one more than waves 4..7 where the pieces
// do not divide by 8
template <class G, int CLS, bool UT>
struct G8Cnt {
  static constexpr int A0 = G::NPA0 / 8 + ((G::NPA0 % 8) && CLS == 0 ? 1 : 0);
  static constexpr int A1 = G::NPA1 / 8 + ((G::NPA1 % 8) && CLS == 0 ? 1 : 0);
  static constexpr int B = 2;
  static constexpr int B1 = B + (UT ? 1 : 0);   // the slab of Ut travels with B1
  static constexpr int ALL = A0 + A1 + B + B1;
};

// MODE: 0 plain product; 1 K-extension with T given (A2 / B2); 2 the adapter inside (Ut / B2: T computed here); 3 the adapter
// inside with T computed by the workgroup's helper waves (gemm8h_kernel): the tile only waits for their image of T
// DV: the launch also computes dVs = A^T T (+ the column sums of A) from the A sub-buffers of its own K loop (cara_gemm_args::er_Tt with
// CARA_EPI_BF16: the dX GEMM of a linear stages that linear's dY as its A operand, whole rows in LDS).  The column tiles of a row panel
// stage the same A: tile column tn takes the K steps t = tn (mod tiles_n), and in such a step the wave row (t / tiles_n) & 1; a wave's
// share = the 16 columns (of the 64 of the K step) of its wave column over all MT rows: five 32-row chunks, each two transposing LDS
// reads of the swizzled sub-buffer (in the phases in which the tile's own fragment reads happen: sub-tile 0 in phase 1, sub-tile 1 in
// phase 3) and one MFMA against that chunk's rows of T^T, one more against (masked) ones for the column sums.  [16 r x 16 k] per wave
// and active K step goes out as ONE 16-byte store per lane into slab (row tile, K step) of the products' slab layout; every wave issues
// the same two store instructions at the end of EVERY K step (beyond num_records where it has nothing to store), so that the counted
// waits stay constants: C::ALL + 2.
template <class G, int EPI, int CLS, int MODE, bool DV = false>
__device__ __forceinline__ void g8_tile(const cara_gemm_args& p, const int tiles_n, const int nwg, const int block, char* smem) {
  constexpr bool EXT = MODE != 0, UT = MODE == 2;
  using C = G8Cnt<G, CLS, UT>;
  constexpr int RT0 = G::RT0, RT1 = G::RT1, RT = G::RT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = xcd_remap(block, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * G::MT, n0 = tn * 256;
  // (operands span < 4 GiB: checked at dispatch)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((unsigned)p.M * (unsigned)p.lda * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((unsigned)p.N * (unsigned)p.ldb * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsEA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(MODE == 1 ? p.A2 : p.A), 0, (int)((unsigned)p.M * 64u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(UT ? p.Ut : p.A), 0, (int)(16u * (unsigned)p.K * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsEB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(EXT ? p.B2 : p.B), 0, (int)((unsigned)p.N * 64u), 0x00020000);

  // ---- staging (LDS-DMA by buffer loads): address = resource base + this lane's offset INSIDE a piece (one VGPR per operand,
  // never changes) + a scalar offset = the piece's first row + the K step: no vector arithmetic per piece.  Pieces that lie wholly
  // beyond M / N re-load the operand's last rows (M, N % 16 == 0: no piece straddles the edge; their outputs are never stored).
  // A piece = 8 rows x 128 B: lane -> row lane >> 3, 16-byte chunk (lane & 7) ^ (row & 7) (the swizzle, on the SOURCE side).
  const int vA = (lane >> 3) * (p.lda * 2) + (((lane & 7) ^ (lane >> 3)) * 16);
  const int vB = (lane >> 3) * (p.ldb * 2) + (((lane & 7) ^ (lane >> 3)) * 16);
  // extension operands: rows of 64 B, a piece = 16 rows x 64 B, chunk (lane & 3) ^ (((row >> 3) & 1) * 3)
  const int vE = (lane >> 2) * 64 + (((lane & 3) ^ (((lane >> 5) & 1) * 3)) * 16);
  // a DMA instruction that moves nothing: a lane offset beyond every resource's num_records (< 2 GiB, checked at dispatch) reads
  // as zeros without a memory request.  The K steps beyond the last one are staged that way, so that EVERY phase of EVERY K step
  // issues the same number of instructions and the counted waits are the same constants from the first phase to the last.
  const int vD = 0x7fffff00;
  // adapter inside: rows 0 .. 15 of Ut [Rp, K] (rank <= 16: the others are zero), two pieces of 8 rows, moved by waves 0 and 1
  const int vU = wave < 2 ? (lane >> 3) * (p.K * 2) + (((lane & 7) ^ (lane >> 3)) * 16) : vD;
  const int sU = wave < 2 ? 8 * wave * (p.K * 2) : 0;
  int sA0[2], sA1[2], sB0[2], sB1[2], sEA[2], sEB[2];   // scalar: byte offset of the first row of this wave's pieces
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int q = wave + 8 * t;
    {   // A sub0: local rows [8q, 8q + 8) of [wave row][16 RT0 rows]
      const int r = 8 * q < G::A0_ROWS ? 8 * q : 0;
      const int w2 = r / (16 * RT0), in = r - w2 * (16 * RT0);
      const int g = m0 + w2 * (16 * RT) + in;
      sA0[t] = (g < p.M ? g : p.M - 8) * (p.lda * 2);
    }
    {   // A sub1
      const int r = 8 * q < G::A1_ROWS ? 8 * q : 0;
      const int w2 = r / (16 * RT1), in = r - w2 * (16 * RT1);
      const int g = m0 + w2 * (16 * RT) + 16 * RT0 + in;
      sA1[t] = (g < p.M ? g : p.M - 8) * (p.lda * 2);
    }
    {   // B sub0 / sub1: local rows of [wave column][32 columns]
      const int r = 8 * q;
      const int g = n0 + (r >> 5) * 64 + (r & 31);
      sB0[t] = (g < p.N ? g : p.N - 8) * (p.ldb * 2);
      sB1[t] = (g + 32 < p.N ? g + 32 : p.N - 8) * (p.ldb * 2);
    }
    sEA[t] = (m0 + 16 * q < p.M ? m0 + 16 * q : p.M - 16) * 64;
    sEB[t] = (n0 + 16 * q < p.N ? n0 + 16 * q : p.N - 16) * 64;
  }
#define G8_DMA(RSRC, VOFF, SOFF, CNT, SUBOFF, BUFX)                                                                         \
  do {                                                                                                                     \
    _Pragma("unroll") for (int t_ = 0; t_ < (CNT); ++t_)                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, (LDS_AS void*)(smem + (BUFX) * G::BUF + (SUBOFF) + (wave + 8 * t_) * 1024), 16, \
                                                 VOFF, SOFF, 0, 0);                                                        \
  } while (0)
  // sub-buffer KIND of K step KT into buffer BUFX.  KT >= nk: nothing to stage (vD) -- except, with the K-extension, the A0 / B0
  // slots of "K step nk", which take the extension's operands T / Vs (64-byte rows, the same instruction counts)
#define G8_ST_A1(KT, BUFX) G8_DMA(rsA, ((KT) < nk ? vA : vD), sA1[t_] + (KT) * 128, C::A1, G::A1_OFF, BUFX)
#define G8_ST_B1(KT, BUFX)                                                                                                  \
  do {                                                                                                                     \
    G8_DMA(rsB, ((KT) < nk ? vB : vD), sB1[t_] + (KT) * 128, C::B, G::B1_OFF, BUFX);                                      \
    if constexpr (UT)                                                                                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (LDS_AS void*)(smem + (wave < 2 ? G::U_OFF + (BUFX) * 2048 + wave * 1024 : G::JUNK_OFF)), 16, \
                                               ((KT) < nk ? vU : vD), sU + (KT) * 128, 0, 0);                              \
  } while (0)
#define G8_ST_A0(KT, BUFX)                                                                                                  \
  do {                                                                                                                     \
    const bool x_ = MODE == 1 && (KT) == nk;                                                                               \
    G8_DMA((x_ ? rsEA : rsA), ((KT) < nk ? vA : (x_ ? vE : vD)), (x_ ? sEA[t_] : sA0[t_] + (KT) * 128), C::A0, G::A0_OFF, BUFX); \
  } while (0)
#define G8_ST_B0(KT, BUFX)                                                                                                  \
  do {                                                                                                                     \
    const bool x_ = EXT && (KT) == nk;                                                                                     \
    G8_DMA((x_ ? rsEB : rsB), ((KT) < nk ? vB : (x_ ? vE : vD)), (x_ ? sEB[t_] : sB0[t_] + (KT) * 128), C::B, G::B0_OFF, BUFX); \
  } while (0)


  // ---- fragment addresses: row (.. + fr) of a sub-buffer, 16-byte chunk (4 kh + fq) ^ (row & 7) ----
  int pa0[2], pa1[2], pb[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int sw = ((kh * 4 + fq) ^ (fr & 7)) * 16;
    pa0[kh] = (wr * 16 * RT0 + fr) * 128 + sw;
    pa1[kh] = (wr * 16 * RT1 + fr) * 128 + sw;
    pb[kh] = (wc * 32 + fr) * 128 + sw;
  }
  // adapter inside: this wave's two tiles of T -- row tile ia of sub-tile 0 and row tile ib of sub-tile 1 of its wave row (the wave
  // columns share a wave row's rows: column wc takes tile wc of each sub-tile, the columns beyond the last tile repeat it unused)
  constexpr int IA_MAX = RT0 - 1, IB_MAX = RT1 - 1;
  const int ia = wc < IA_MAX ? wc : IA_MAX, ib = wc < IB_MAX ? wc : IB_MAX;
  int pta[2], ptb[2], pu[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int sw = ((kh * 4 + fq) ^ (fr & 7)) * 16;
    pta[kh] = G::A0_OFF + (wr * 16 * RT0 + ia * 16 + fr) * 128 + sw;
    ptb[kh] = G::A1_OFF + (wr * 16 * RT1 + ib * 16 + fr) * 128 + sw;
    pu[kh] = G::U_OFF + fr * 128 + sw;
  }
  // ---- DV: this lane's rows of T^T per 32-row chunk (chunks 0 .. NC0 - 1 of sub-tile 0, the rest of sub-tile 1), the validity of
  // those rows (the last row tile re-stages the operand's last rows beyond M: T reads as zero there, the ones as zero too), the
  // transposing-read offsets inside a sub-buffer and the two store resources ----
  constexpr int NC0 = G::A0_ROWS / 32, NC1 = G::A1_ROWS / 32, NC = NC0 + NC1;
  static_assert(!DV || (G::A0_ROWS % 32 == 0 && G::A1_ROWS % 32 == 0 && (16 * RT0) % 8 == 0 && (16 * RT1) % 8 == 0), "DV: whole 32-row chunks of 8-row runs");
  bf16x8 tfr[DV ? NC : 1];
  unsigned dv_valid = 0;
  unsigned dv_lo = 0, dv_hi = 0;
  __amdgpu_buffer_rsrc_t rsDV = rsA;
  int vDV = vD, vDC = vD, dv_cs_base = 0;
  bool dv_act = false;
  f32x4 accD = {0.f, 0.f, 0.f, 0.f};
  float accC = 0.f;
  s16x4 dvl[DV ? (NC0 > NC1 ? NC0 : NC1) : 1], dvh[DV ? (NC0 > NC1 ? NC0 : NC1) : 1];
  const int tiles_m = nwg / tiles_n;
  if constexpr (DV) {
    const bf16* Tt = static_cast<const bf16*>(p.er_Tt) + (size_t)fr * p.er_ldg;
    const bf16x8 z8 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int r0 = (c < NC0 ? 32 * c : 32 * (c - NC0)) + 8 * fq;   // local row of the sub-buffer
      const int rows = c < NC0 ? 16 * RT0 : 16 * RT1;
      const int w2 = r0 / rows, in = r0 - w2 * rows;
      const int g = m0 + w2 * (16 * RT) + (c < NC0 ? 0 : 16 * RT0) + in;
      const bool ok = g < p.M;   // (M % 16 == 0: the eight rows are valid together)
      tfr[c] = ok ? *reinterpret_cast<const bf16x8*>(Tt + g) : z8;
      dv_valid |= ok ? (1u << c) : 0u;
    }
    // rows 8 fq + (fr >> 2) (and + 4) of a chunk, the 8 bytes at column 16 wc + 4 (fr & 3) of the K step's 64: logical 16-byte chunk
    // 2 wc + ((fr & 3) >> 1), XOR the row's swizzle key (row & 7)
    const int rl = 8 * fq + (fr >> 2), ch = 2 * wc + ((fr & 3) >> 1);
    dv_lo = (unsigned)(rl * 128 + ((ch ^ (rl & 7)) << 4) + (fr & 1) * 8);
    dv_hi = (unsigned)((rl + 4) * 128 + ((ch ^ ((rl + 4) & 7)) << 4) + (fr & 1) * 8);
    const int colblocks = p.K >> 6;
    const size_t slab_floats = (size_t)tiles_m * colblocks * (64 * 16), cs_off = (size_t)tiles_m * colblocks * (64 * 32);
    // ONE resource over the slabs and the column sums behind them (no column sums: its bound ends at the slabs, those stores are dropped)
    rsDV = __builtin_amdgcn_make_buffer_rsrc(p.er_slabs_v, 0, (int)(p.er_colsum ? (cs_off + (size_t)tiles_m * colblocks * 64) * 4 : slab_floats * 4), 0x00020000);
    dv_cs_base = (int)(cs_off * 4);
    vDV = ((16 * wc + fr) * 16 + 4 * fq) * 4;
    vDC = fq == 0 ? (16 * wc + fr) * 4 : vD;
    // the loads above are complete before the loop (so that no wait of the compiler's for them lands inside it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(tfr[c]));
  }
  bf16x8 ta[2], u[2];
  f32x4 accT[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 a[RT][2], b[4][2];
  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define G8_LD(SUBOFF, IDX, PTR, BUFX) (*reinterpret_cast<const bf16x8*>(smem + (BUFX) * G::BUF + (SUBOFF) + (IDX) * 2048 + (PTR)))
#define G8_RD_A0(BUFX)                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < RT0; ++i_) {                                    \
    a[i_][0] = G8_LD(G::A0_OFF, i_, pa0[0], BUFX);                                        \
    a[i_][1] = G8_LD(G::A0_OFF, i_, pa0[1], BUFX);                                        \
  }
#define G8_RD_A1(BUFX)                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < RT1; ++i_) {                                    \
    a[RT0 + i_][0] = G8_LD(G::A1_OFF, i_, pa1[0], BUFX);                                  \
    a[RT0 + i_][1] = G8_LD(G::A1_OFF, i_, pa1[1], BUFX);                                  \
  }
#define G8_RD_B0(BUFX)                                                                    \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                      \
    b[j_][0] = G8_LD(G::B0_OFF, j_, pb[0], BUFX);                                         \
    b[j_][1] = G8_LD(G::B0_OFF, j_, pb[1], BUFX);                                         \
  }
#define G8_RD_B1(BUFX)                                                                    \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                      \
    b[2 + j_][0] = G8_LD(G::B1_OFF, j_, pb[0], BUFX);                                     \
    b[2 + j_][1] = G8_LD(G::B1_OFF, j_, pb[1], BUFX);                                     \
  }
  // TW = 0 / 1: with the adapter inside, the phase also adds the two 32-deep halves of this K step to T tile TW (first and last in
  // the cluster: the second depends on the first)
#define G8_MMA(I0, I1, J0, J1, TW)                                                                               \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    if constexpr (UT && (TW) >= 0) accT[(TW) < 0 ? 0 : (TW)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta[0], u[0], accT[(TW) < 0 ? 0 : (TW)], 0, 0, 0); \
    _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                                          \
        _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_)                                                   \
            _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_)                                               \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i_][kh_], b[j_][kh_], acc[i_][j_], 0, 0, 0); \
    if constexpr (UT && (TW) >= 0) accT[(TW) < 0 ? 0 : (TW)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta[1], u[1], accT[(TW) < 0 ? 0 : (TW)], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)
  // one phase behind its reads and DMA issue: counted wait (one sub-buffer of each kind stays in flight), barrier, this
  // quadrant's MFMAs, barrier
  constexpr int WAITN = C::ALL + (DV ? 2 : 0);
  // DV: the transposing reads of sub-tile SUB of K step T in buffer BUFX (wave-uniform branch; inline asm: behind the builtin the compiler
  // puts s_waitcnt vmcnt(0), it cannot tell the read from the LDS-DMA in flight), and their MFMAs (after the phase's lgkmcnt(0))
#define G8_DV_RD(SUB, BUFX)                                                                                                         \
  do {                                                                                                                             \
    if (dv_act) {                                                                                                                  \
      const unsigned lb_ = (unsigned)(uintptr_t)(LDS_AS const char*)(smem + (BUFX) * G::BUF + ((SUB) ? G::A1_OFF : G::A0_OFF));    \
      _Pragma("unroll") for (int c_ = 0; c_ < ((SUB) ? NC1 : NC0); ++c_) {                                                         \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dvl[c_]) : "v"(lb_ + dv_lo + c_ * 4096) : "memory");                       \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dvh[c_]) : "v"(lb_ + dv_hi + c_ * 4096) : "memory");                       \
      }                                                                                                                            \
    }                                                                                                                              \
  } while (0)
#define G8_DV_MMA(SUB)                                                                                                              \
  do {                                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                             \
    if (dv_act) {                                                                                                                  \
      _Pragma("unroll") for (int c_ = 0; c_ < ((SUB) ? NC1 : NC0); ++c_) {                                                         \
        /* K = 16 MFMAs on the two transposing reads as they are (rows 8 fq .. + 3 and + 4 .. + 7 of the chunk: the halves of tfr) */ \
        const bf16x8 t_ = tfr[((SUB) ? NC0 : 0) + c_];                                                                             \
        const bf16x4 tl_ = {t_[0], t_[1], t_[2], t_[3]}, th_ = {t_[4], t_[5], t_[6], t_[7]};                                       \
        const bf16x4 l4_ = __builtin_bit_cast(bf16x4, dvl[c_]), h4_ = __builtin_bit_cast(bf16x4, dvh[c_]);                         \
        accD = mfma_16x16x16(tl_, l4_, accD);                                                                                      \
        accD = mfma_16x16x16(th_, h4_, accD);                                                                                      \
        /* column sums: this lane's eight rows of column 16 wc + fr (rows >= M: not counted) */                                    \
        const float cs_ = (((float)l4_[0] + (float)l4_[1]) + ((float)l4_[2] + (float)l4_[3])) +                                    \
                          (((float)h4_[0] + (float)h4_[1]) + ((float)h4_[2] + (float)h4_[3]));                                    \
        accC += ((dv_valid >> (((SUB) ? NC0 : 0) + c_)) & 1u) ? cs_ : 0.f;                                                         \
      }                                                                                                                            \
    }                                                                                                                              \
  } while (0)
#define G8_SYNC_MMA(I0, I1, J0, J1, TW, DVS) \
  do {                                  \
    g8_vmcnt<WAITN>();                  \
    g8_barrier();                       \
    g8_lgkm0();                         \
    if constexpr (DV && (DVS) >= 0) G8_DV_MMA(DVS); \
    G8_MMA(I0, I1, J0, J1, TW);         \
    g8_barrier();                       \
  } while (0)
  // K step T in buffer CUR
  // (adapter inside: the slab of Ut of this K step landed with B1 -- read in phase 2; T tile 0 = a row tile of sub-tile 0, its A
  // fragments re-read in phase 1, multiplied in phase 2; T tile 1 of sub-tile 1: read in phase 3, multiplied in phase 4)
#define G8_TILE(T, CUR)                                                                                  \
  do {                                                                                                   \
    if constexpr (DV) {                                                                                  \
      const int q_ = (T) / tiles_n;                                                                      \
      dv_act = (T) - q_ * tiles_n == tn && (q_ & 1) == wr;                                               \
    }                                                                                                    \
    G8_RD_A0(CUR) G8_RD_B0(CUR)                                                                          \
    if constexpr (DV) G8_DV_RD(0, CUR);                                                                  \
    if constexpr (UT) {                                                                                  \
      ta[0] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * G::BUF + pta[0]);                          \
      ta[1] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * G::BUF + pta[1]);                          \
    }                                                                                                    \
    G8_ST_B1((T) + 1, (CUR) ^ 1);                                                                        \
    G8_SYNC_MMA(0, RT0, 0, 2, -1, 0);                                                                    \
    G8_RD_B1(CUR)                                                                                        \
    if constexpr (UT) {                                                                                  \
      u[0] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * 2048 + pu[0]);                              \
      u[1] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * 2048 + pu[1]);                              \
    }                                                                                                    \
    G8_ST_A1((T) + 1, (CUR) ^ 1);                                                                        \
    G8_SYNC_MMA(0, RT0, 2, 4, 0, -1);                                                                    \
    G8_RD_A1(CUR)                                                                                        \
    if constexpr (DV) G8_DV_RD(1, CUR);                                                                  \
    if constexpr (UT) {                                                                                  \
      ta[0] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * G::BUF + ptb[0]);                          \
      ta[1] = *reinterpret_cast<const bf16x8*>(smem + (CUR) * G::BUF + ptb[1]);                          \
    }                                                                                                    \
    G8_ST_A0((T) + 2, CUR);                                                                              \
    G8_SYNC_MMA(RT0, RT, 2, 4, -1, 1);                                                                   \
    G8_ST_B0((T) + 2, CUR);                                                                              \
    G8_SYNC_MMA(RT0, RT, 0, 2, 1, -1);                                                                   \
    if constexpr (DV) {                                                                                  \
      /* this K step's [16 r x 16 k] sums: row 4 fq + reg = r, column fr = k -> slab (row tile, K step), 16 bytes per lane */ \
      const int so_ = (tm * (p.K >> 6) + (T)) * (64 * 16 * 4), sc_ = dv_cs_base + (tm * (p.K >> 6) + (T)) * (64 * 4);   \
      float c_ = accC;                                                                                   \
      c_ += __shfl_xor(c_, 16, 64);                                                                      \
      c_ += __shfl_xor(c_, 32, 64);                                                                      \
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, accD), rsDV, dv_act ? vDV : vD, dv_act ? so_ : 0, 0); \
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c_), rsDV, dv_act ? vDC : vD, dv_act ? sc_ : 0, 0); \
      accD = f32x4{0.f, 0.f, 0.f, 0.f};                                                                  \
      accC = 0.f;                                                                                        \
    }                                                                                                    \
  } while (0)

  const int nk = p.K >> 6;   // >= 2 (checked at dispatch)
  // prologue: K step 0 whole, A0 and B0 of K step 1
  G8_ST_A0(0, 0);
  G8_ST_B0(0, 0);
  G8_ST_B1(0, 0);
  G8_ST_A1(0, 0);
  G8_ST_A0(1, 1);
  G8_ST_B0(1, 1);
  if constexpr (DV) {   // (the two store instructions "K step -1" would have issued here: the waits count them from the first phase on)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, accD), rsDV, vD, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0u, rsDV, vD, 0, 0);
  }
  g8_vmcnt<WAITN>();
  g8_barrier();
  if (wr == 1) g8_barrier();   // the second wave row runs one barrier behind the first
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    G8_TILE(t, 0);
    G8_TILE(t + 1, 1);
  }
  if (t < nk) G8_TILE(t, 0);
  if (wr == 0) g8_barrier();
  if constexpr (MODE == 3) __syncthreads();   // the helper waves have written T into the extension's A image
  if constexpr (UT) {
    // T (fp32, 16 x 16 per tile: row 4 fq + reg, column fr) -> bf16 -> the extension's A image (64-byte rows in natural row order
    // in the A0 region of the buffer of "K step nk": nothing has read or written it since that K step's phase 3); columns 16 .. 31
    // are zero (rank <= 16).  The owners: wave column wc < RT0 for row tile wc, wc < RT1 for row tile RT0 + wc.  The workgroups
    // of column 0 also write T [M, 32] and Tt [32, ldt] for the backward.
    char* ea = smem + (nk & 1) * G::BUF + G::A0_OFF;
    bf16* Tg = static_cast<bf16*>(p.T_out);
    bf16* Ttg = static_cast<bf16*>(p.Tt_out);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const bool own = w == 0 ? wc < RT0 : wc < RT1;   // wave-uniform
      if (!own) continue;
      const int row0 = wr * 16 * RT + (w == 0 ? wc : RT0 + wc) * 16 + fq * 4;
      const f32x4 av = accT[w];
      const bf16x4 tv = {(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + r, sz = ((row >> 3) & 1) * 3;
        *reinterpret_cast<bf16*>(ea + row * 64 + (((fr >> 3) ^ sz) << 4) + (fr & 7) * 2) = tv[r];
        *reinterpret_cast<bf16*>(ea + row * 64 + (((2 + (fr >> 3)) ^ sz) << 4) + (fr & 7) * 2) = (bf16)0.f;
      }
      if (tn == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + row0 + r < p.M) {
            Tg[(size_t)(m0 + row0 + r) * 32 + fr] = tv[r];
            Tg[(size_t)(m0 + row0 + r) * 32 + 16 + fr] = (bf16)0.f;
          }
        if (Ttg && m0 + row0 + 4 <= p.M) {   // (M % 16 == 0: a group of four rows is inside or outside)
          *reinterpret_cast<bf16x4*>(Ttg + (size_t)fr * p.ldt + m0 + row0) = tv;
          *reinterpret_cast<bf16x4*>(Ttg + (size_t)(16 + fr) * p.ldt + m0 + row0) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
      }
    }
    __syncthreads();
  }
  if constexpr (EXT) {
    // [T | Vs]: 64-byte rows in natural row order in the A0 / B0 regions of the buffer of "K step nk" (landed: the waits of the
    // last K step's phases 3 and 4, a barrier ago at least), one 32-deep step
    const char* ea = smem + (nk & 1) * G::BUF + G::A0_OFF;
    const char* eb = smem + (nk & 1) * G::BUF + G::B0_OFF;
#pragma unroll
    for (int i = 0; i < RT; ++i) a[i][0] = *reinterpret_cast<const bf16x8*>(ea + (wr * 16 * RT + i * 16 + fr) * 64 + ((fq ^ (((fr >> 3) & 1) * 3)) << 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j][0] = *reinterpret_cast<const bf16x8*>(eb + (wc * 64 + j * 16 + fr) * 64 + ((fq ^ (((fr >> 3) & 1) * 3)) << 4));
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
  }
  g8_vmcnt<0>();     // (the zero-fill DMA of the K steps beyond the last has landed too)
  __syncthreads();   // every wave is through with the buffers: the epilogue's staging images go there

  // ---- epilogue: the paths of gemm_epilogue.h on this wave's (16 RT) x 64 outputs ----
  constexpr int WAVE_STG = 16 * 64 * 4 > EPI_FAST_WAVE_BYTES ? 16 * 64 * 4 : EPI_FAST_WAVE_BYTES;
  char* wstg = smem + wave * WAVE_STG;
  const int mw = m0 + wr * (RT * 16), nw = n0 + wc * 64;
  const bool interior = mw + RT * 16 <= p.M && nw + 64 <= p.N && (p.ldc & 7) == 0;   // wave-uniform
  if constexpr (EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU) {
    if (interior) {
      epilogue_fast_bf16_rt<EPI, RT>(p, acc, wstg, mw, nw, lane, 0);
      return;
    }
  }
  float* stg = reinterpret_cast<float*>(wstg);
  if constexpr (EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU) {
    const bool ok = interior && (EPI == CARA_EPI_DGELU || !p.rowscale || p.rows_per_sample >= RT * 16) && (!p.bias || (nw & 3) == 0);
    if (ok) {
      epilogue_interior_aux<EPI, RT, 1>(p, acc, stg, mw, nw, lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * 64 + j * 16 + fr] = acc[i][j][r];
    asm volatile("" ::: "memory");
    epilogue_rows<EPI, 16>(p, stg, mw + i * 16, nw, lane, 0);
    asm volatile("" ::: "memory");
  }
#undef G8_DMA
#undef G8_ST_A0
#undef G8_ST_A1
#undef G8_ST_B0
#undef G8_ST_B1
#undef G8_LD
#undef G8_RD_A0
#undef G8_RD_A1
#undef G8_RD_B0
#undef G8_RD_B1
#undef G8_MMA
#undef G8_SYNC_MMA
#undef G8_DV_RD
#undef G8_DV_MMA
#undef G8_TILE
}

// stagger > 0 (products of several rounds of tiles: N = 4 dim / 3 dim): the workgroups of the FIRST round start (blockIdx % 4) x
// stagger x 64 clocks apart.  One tile per CU makes every CU end its tile at the same moment: all epilogues of a round hit HBM
// together (155 MB per launch for fc1 forward: 31 us at 5 TB/s, four bursts with the chip's matrix pipes idle) and the next round
// starts in lockstep again.  A quarter-period offset between neighbouring CUs, paid once, lets one group's stores run under the
// other groups' K loops for the rest of the launch.
template <class G, int EPI, int MODE, bool DV = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (stagger > 0 && blockIdx.x < 256) {
    const int n = (int)((blockIdx.x >> 3) & 3) * stagger;   // (blocks b, b + 8 share an XCD: the offset varies inside an XCD)
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  constexpr bool TWO_CLASSES = (G::NPA0 % 8) != 0 || (G::NPA1 % 8) != 0;
  if constexpr (TWO_CLASSES) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) g8_tile<G, EPI, 0, MODE, DV>(p, tiles_n, nwg, blockIdx.x, smem);
    else g8_tile<G, EPI, 1, MODE, DV>(p, tiles_n, nwg, blockIdx.x, smem);
  } else {
    g8_tile<G, EPI, 0, MODE, DV>(p, tiles_n, nwg, blockIdx.x, smem);
  }
}

// The same grid with the blocks of a pair of transposed skinny products (rank <= 16: one r-tile) behind the tiles: a 512-thread
// workgroup runs two of them side by side (the device code of tskinny_kernel with its three-deep ring per wave: 8 waves are all
// a CU holds here).  They start on the CUs the tiles leave free and spread over the others as the tiles finish.
template <class G, int EPI, int MODE, bool COLSUM, bool DV = false>
__global__ __launch_bounds__(512, 2) void gemm8_ts_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const TsProblem t0,
                                                          const TsProblem t1, const int ldg, const int Mts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (b >= nwg) {
    constexpr int TSB = TsRing<1, 3>::BLOCK_BYTES;
    static_assert(2 * TSB <= 160 * 1024, "two tskinny blocks in one workgroup's LDS");   // (the launch asks for max(G::LDS, 2 TSB))
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int nts = t0.nblk + t1.nblk;
    const int blk = 2 * (b - nwg) + sub;
    tskinny_body<1, COLSUM, 3>(t0, t1, ldg, Mts, blk < nts ? blk : nts - 1, smem + sub * TSB, threadIdx.x & 255, blk < nts);
    return;
  }
  constexpr bool TWO_CLASSES = (G::NPA0 % 8) != 0 || (G::NPA1 % 8) != 0;
  if constexpr (TWO_CLASSES) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) g8_tile<G, EPI, 0, MODE, DV>(p, tiles_n, nwg, b, smem);
    else g8_tile<G, EPI, 1, MODE, DV>(p, tiles_n, nwg, b, smem);
  } else {
    g8_tile<G, EPI, 0, MODE, DV>(p, tiles_n, nwg, b, smem);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Helper waves (gemm8h_kernel): a THIRD wave per SIMD -- the tile needs 165 of the 168 VGPRs three waves per SIMD may have -- that
// runs the barrier sequence of a tile wave of row 0 and, between the barriers, does what would otherwise wait for the tile's end:
//  * the transposed skinny products the launch carries (dU = X^T G', dVs = dY^T T of cara_gemm_with_tskinny_r): the K loop of the
//    tile is bound by MFMA issue and by the CU's L2 -> LDS path while HBM idles; the products are HBM streams.  One step (32 rows x
//    64 columns of X + 16 x 32 of Gt: 5 LDS-DMA instructions into one of the wave's two private 5-KiB stages) is issued per K step
//    and consumed almost two K steps later (a transposing-read MFMA step as in tskinny_body.h), with the wave's OWN vmcnt -- the
//    tile waves' counted waits never see it.  Work unit = the steps of ONE WAVE of a tskinny block (block b, wave w: steps
//    s_begin + w, + 4, ...): a unit's sums go to a slab of its own (4 slabs per block: cara_ts_reduce::wave_slabs), no combine
//    through LDS, no barrier of its own.  Helper wave (tile i, h) takes the units 4 i + h, + 4 nwg, ...; what is left when the tile's
//    K loop ends runs behind the last barrier.
//  * with the adapter inside (MODE 3): T = A Ut^T for the tile's rows from the A sub-buffers the tile waves stage anyway (read in
//    the phases in which a tile wave of row 0 may read them) and a slab of Ut the helper waves stage themselves two K steps ahead:
//    2 RT tiles of 16 x 16 over four waves, 5 MFMAs per wave and K step on the matrix pipe's account instead of +4 in EVERY tile
//    wave, and no registers of the tile waves.
// ---------------------------------------------------------------------------------------------------------------
constexpr int H8_STAGE = TsRing<1, 1>::STAGE;   // 5 KiB: X 32 x 64 | Gt 16 x 32 (bf16)
template <class G>
struct H8 {
  static constexpr int U3_OFF = G::LDS;                       // three 2-KiB slabs of Ut (staged two K steps ahead)
  static constexpr int RING_OFF = U3_OFF + 3 * 2048;
  static constexpr int LDS = RING_OFF + 4 * 2 * H8_STAGE;     // MT = 160: 111616 + 6144 + 40960 = 158720 B of 160 KiB
  static_assert(LDS <= 160 * 1024, "the tile, the Ut slabs and four two-stage rings share a CU's LDS");
};

__device__ __forceinline__ void h8_wait(const int n) {   // n: wave-uniform, the LDS-DMA instructions that may stay in flight
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;   // 6
  }
}

// the unit a helper wave is streaming: one wave's steps of one tskinny block
struct H8Unit {
  __amdgpu_buffer_rsrc_t rsX, rsG;   // X [Mts, ldx] and rows 0 .. 15 of Gt [Rp, ldg], bounds = their sizes (rows beyond read as zeros)
  float* slab; float* cs;            // this unit's 64 x 16 slab and (or null) its 64 column sums
  int ldx2, col2;                    // bytes per row of X; byte offset of the unit's 64 columns
  int next, end;                     // steps next, next + 4, ... < end are left
};

template <class G, int MODE, bool COLSUM>
__device__ __forceinline__ void g8_helper(const cara_gemm_args& p, const int tiles_n, const int nwg, const int block, const TsProblem& t0,
                                          const TsProblem& t1, const int ldg, const int Mts, char* smem) {
  constexpr bool UT = MODE == 3;
  constexpr int RT0 = G::RT0, RT1 = G::RT1, RT = G::RT;
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - 8;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = xcd_remap(block, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * G::MT;
  const int nk = p.K >> 6;

  // ---- T = A Ut^T: this wave's tiles (wave row, row tile) = h, h + 4, h + 8 of the 2 RT; sub-tile 0 tiles are read in phase 1,
  // sub-tile 1 tiles in phase 3 (where a tile wave of row 0 reads those sub-buffers) ----
  f32x4 accT[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  int taddr[3][2], trow[3];
  bool tsub1[3], tvalid[3];
  int pu[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int id = h + 4 * k;
    tvalid[k] = UT && id < 2 * RT;
    const int w2 = (id < 2 * RT ? id : 0) / RT, i = (id < 2 * RT ? id : 0) - w2 * RT;
    tsub1[k] = i >= RT0;
    trow[k] = w2 * 16 * RT + i * 16;
    const int base = tsub1[k] ? G::A1_OFF + (w2 * 16 * RT1 + (i - RT0) * 16 + fr) * 128 : G::A0_OFF + (w2 * 16 * RT0 + i * 16 + fr) * 128;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) taddr[k][kh] = base + (((kh * 4 + fq) ^ (fr & 7)) * 16);
  }
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) pu[kh] = H8<G>::U3_OFF + fr * 128 + (((kh * 4 + fq) ^ (fr & 7)) * 16);
  bf16x8 u[2];
  // rows 8 h .. 8 h + 7 of Ut (h < 2: rank <= 16), 128 B of K step kt -> slab kt % 3
  const bf16* Ut = static_cast<const bf16*>(p.Ut);
  auto stage_u = [&](int kt) {
    if (UT && h < 2 && kt < nk)
      glds16(Ut + (size_t)(8 * h + (lane >> 3)) * p.K + kt * 64 + (((lane & 7) ^ (lane >> 3)) * 8), smem + H8<G>::U3_OFF + (kt % 3) * 2048 + h * 1024);
    return (UT && h < 2 && kt < nk) ? 1 : 0;
  };
  auto t_part = [&](int kt, bool second) {   // the MFMAs of this wave's T tiles of sub-tile 0 (second = false) or 1
    if constexpr (UT) {
      const char* buf = smem + (kt & 1) * G::BUF;
      if (!second) {
        u[0] = *reinterpret_cast<const bf16x8*>(smem + (kt % 3) * 2048 + pu[0]);
        u[1] = *reinterpret_cast<const bf16x8*>(smem + (kt % 3) * 2048 + pu[1]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (tvalid[k] && tsub1[k] == second) {   // wave-uniform
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(buf + taddr[k][0]);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(buf + taddr[k][1]);
          accT[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, u[0], accT[k], 0, 0, 0);
          accT[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, u[1], accT[k], 0, 0, 0);
        }
      g8_lgkm0();   // (the sub-buffer is restaged two phases on: the reads are done before this wave's next barrier)
    }
  };

  // ---- the riding products: units 4 block + h, + 4 nwg, ... of the 4 (t0.nblk + t1.nblk).  A step = 5 LDS-DMA instructions by
  // buffer loads (the rows beyond Mts are beyond num_records and read as zeros: no clamping, and the column sums need no mask);
  // its sums = 4 MFMAs (X^T G') + 4 more against an all-ones operand for the column sums of X (instead of 100 vector
  // instructions): a helper wave must stay well inside a barrier interval of the tile (~230 cycles), every cycle it is late
  // there all twelve waves wait. ----
  const int nunits = 4 * (t0.nblk + t1.nblk);
  const int steps = (Mts + 31) / 32;
  int ui = 4 * block + h;
  H8Unit cur;
  cur.next = cur.end = 0;
  cur.slab = nullptr; cur.cs = nullptr; cur.ldx2 = 0; cur.col2 = 0;
  cur.rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0, 0x00020000);
  cur.rsG = cur.rsX;
  bool have = false;   // a unit is open
  auto open_unit = [&]() {   // the unit ui (ui < nunits)
    const int b4 = ui >> 2, w = ui & 3;
    const bool second = b4 >= t0.nblk;
    const TsProblem& P = second ? t1 : t0;
    const int bid = second ? b4 - t0.nblk : b4;
    const int colblocks = P.K1 / TS_COLS;
    const int cb = bid % colblocks, chunk = bid / colblocks;
    const int s_begin = (int)((long)steps * chunk / P.nchunks), s_end = (int)((long)steps * (chunk + 1) / P.nchunks);
    const int slab_id = (chunk * 4 + w) * colblocks + cb;   // [4 nchunks][colblocks] slabs of 64 x 16
    cur.rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(P.X), 0, (int)((unsigned)Mts * (unsigned)P.ldx * 2u), 0x00020000);
    cur.rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(P.Gt), 0, (int)(16u * (unsigned)ldg * 2u), 0x00020000);
    cur.ldx2 = P.ldx * 2;
    cur.col2 = cb * TS_COLS * 2;
    cur.slab = P.slabs + (size_t)slab_id * TS_COLS * 16;
    // (the column sums sit behind 4 nblk slabs of the FULL width Rp = 32, where cara_tskinny_reduce_many looks for them)
    cur.cs = (COLSUM && P.cs_slabs) ? P.slabs + (size_t)4 * P.nblk * TS_COLS * 32 + (size_t)slab_id * TS_COLS : nullptr;
    cur.next = s_begin + w;
    cur.end = s_end;
    have = true;
  };
  f32x4 acc[4], accs[4];   // X^T G' (row r = 4 fq + reg, column it * 16 + fr) and, every row the same, the column sums of X
  auto reset_acc = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      accs[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  reset_acc();
  char* ring = smem + H8<G>::RING_OFF + h * 2 * H8_STAGE;
  // per stage: in flight / landed (a step's first row, -1: nothing) and, if it is its unit's last step, where the unit's sums go
  int st_row[2] = {-1, -1};
  float* st_slab[2] = {nullptr, nullptr};
  float* st_cs[2] = {nullptr, nullptr};
  bool st_last[2] = {false, false};
  auto store_sums = [&](float* slab, float* cs, const bool zeros) {
    // C layout of a 16 x 16 tile: row (= r) = 4 fq + reg, column (= i) = 16 it + fr  ->  slab[i][r .. r + 3]
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(slab + (size_t)(it * 16 + fr) * 16 + fq * 4) = zeros ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[it];
    if constexpr (COLSUM) {
      if (cs && fq == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) cs[it * 16 + fr] = zeros ? 0.f : accs[it][0];
      }
    }
  };
  // this lane's part of a step's addresses: X piece q = rows 8 q + (lane >> 3), 16-byte chunk (lane & 7) ^ q of the 64 columns;
  // Gt piece: row lane >> 2 of the 16, 16-byte chunk lane & 3 of the step's 32 m
  const int xr = lane >> 3;
  const int vg = (lane >> 2) * (ldg * 2) + (lane & 3) * 16;
  // issue the next step of the stream into stage sg (free); returns the LDS-DMA instructions issued (5 or 0)
  auto issue = [&](int sg) {
    for (;;) {
      if (!have) {
        if (ui >= nunits) return 0;
        open_unit();
        ui += 4 * nwg;
        if (cur.next >= cur.end) {   // an empty unit still owns a slab (the reduction adds every slab): zeros, straight from here
          store_sums(cur.slab, cur.cs, true);
          have = false;
          continue;
        }
      }
      char* stg = ring + sg * H8_STAGE;
      const int row0 = cur.next * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cur.rsX, (LDS_AS void*)(stg + q * 1024), 16, (row0 + 8 * q + xr) * cur.ldx2 + (((lane & 7) ^ q) * 16), cur.col2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cur.rsG, (LDS_AS void*)(stg + TsRing<1, 1>::X_BYTES), 16, vg, row0 * 2, 0, 0);
      st_row[sg] = row0;
      cur.next += 4;
      st_last[sg] = cur.next >= cur.end;
      st_slab[sg] = cur.slab;
      st_cs[sg] = cur.cs;
      if (st_last[sg]) have = false;
      return 5;
    }
  };
  // a landed step in two halves: its LDS reads (after which the stage is free again) and, from registers, its MFMAs
  bf16x8 ga;
  s16x4 lo[4], hi[4];
  bool rd_valid = false, rd_last = false;
  float *rd_slab = nullptr, *rd_cs = nullptr;
  auto step_reads = [&](int sg) {
    rd_valid = st_row[sg] >= 0;
    if (!rd_valid) return;
    const char* sx = ring + sg * H8_STAGE;
    ga = *reinterpret_cast<const bf16x8*>(sx + TsRing<1, 1>::X_BYTES + fr * 64 + fq * 16);
    // The transposing reads as inline asm: behind the builtin the compiler's wait-count pass puts s_waitcnt vmcnt(0) (it cannot tell
    // the read from the LDS-DMA in flight into the OTHER stage) -- a whole HBM latency per K step for all twelve waves.  What has
    // landed is decided by this wave's counted wait alone.
    const unsigned lbase = (unsigned)(uintptr_t)(LDS_AS const char*)(sx + (fq * 8 + (fr >> 2)) * 128);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const unsigned toff = (unsigned)((((it * 2 + ((fr & 3) >> 1)) ^ fq) << 4) | ((fr & 1) << 3));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[it]) : "v"(lbase + toff) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi[it]) : "v"(lbase + toff) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // (register-only instructions must not move above the wait)
    rd_last = st_last[sg];
    rd_slab = st_slab[sg];
    rd_cs = st_cs[sg];
    st_row[sg] = -1;   // the stage is free
  };
  auto step_mfma = [&]() {
    if (!rd_valid) return;
    rd_valid = false;
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo[it]), h4 = __builtin_bit_cast(bf16x4, hi[it]);
      const bf16x8 b = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, b, acc[it], 0, 0, 0);
      if constexpr (COLSUM) {
        if (rd_cs) accs[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b, accs[it], 0, 0, 0);
      }
    }
    if (rd_last) {
      store_sums(rd_slab, rd_cs, false);
      reset_acc();
    }
  };

  // ---- the tile's barrier sequence (a tile wave of row 0: prologue barrier, 8 per K step, one more, [MODE 3: T image], the last).
  // Per K step: DMA issue | T tiles of sub-tile 0 | MFMAs of the step read at the end of the previous K step | - | T tiles of
  // sub-tile 1 | - | - | wait for the step issued a K step ago (and the slab of Ut of the next K step), its LDS reads ----
  int n_u = stage_u(0);
  n_u += stage_u(1);
  h8_wait(0);
  g8_barrier();   // (prologue)
  for (int kt = 0; kt < nk; ++kt) {
    const int sg = kt & 1;
    int issued = issue(sg);          // (its last step was read at the end of K step kt - 1: free)
    issued += stage_u(kt + 2);
    g8_barrier();
    t_part(kt, false);
    g8_barrier();
    step_mfma();
    g8_barrier(); g8_barrier();
    t_part(kt, true);                // phase 3: sub-tile 1 has landed for every wave
    g8_barrier(); g8_barrier(); g8_barrier();
    h8_wait(issued);                 // everything issued before this K step has landed
    step_reads(sg ^ 1);
    g8_barrier();
  }
  g8_barrier();   // (the tile waves of row 0 wait here for row 1)
  if constexpr (UT) {
    // T (row 4 fq + reg, column fr of each 16 x 16 tile) -> bf16 -> the extension's A image: 64-byte rows in natural row order in
    // the A0 region of the buffer of "K step nk" (free since that buffer's phase 3), columns 16 .. 31 zero; column-0 workgroups
    // write T [M, 32] and Tt [32, ldt] for the backward
    char* ea = smem + (nk & 1) * G::BUF + G::A0_OFF;
    bf16* Tg = static_cast<bf16*>(p.T_out);
    bf16* Ttg = static_cast<bf16*>(p.Tt_out);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!tvalid[k]) continue;
      const int row0 = trow[k] + fq * 4;
      const f32x4 av = accT[k];
      const bf16x4 tv = {(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + r, sz = ((row >> 3) & 1) * 3;
        *reinterpret_cast<bf16*>(ea + row * 64 + (((fr >> 3) ^ sz) << 4) + (fr & 7) * 2) = tv[r];
        *reinterpret_cast<bf16*>(ea + row * 64 + (((2 + (fr >> 3)) ^ sz) << 4) + (fr & 7) * 2) = (bf16)0.f;
      }
      if (tn == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + row0 + r < p.M) {
            Tg[(size_t)(m0 + row0 + r) * 32 + fr] = tv[r];
            Tg[(size_t)(m0 + row0 + r) * 32 + 16 + fr] = (bf16)0.f;
          }
        if (Ttg && m0 + row0 + 4 <= p.M) {
          *reinterpret_cast<bf16x4*>(Ttg + (size_t)fr * p.ldt + m0 + row0) = tv;
          *reinterpret_cast<bf16x4*>(Ttg + (size_t)(16 + fr) * p.ldt + m0 + row0) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_barrier();   // (MODE 3: the tile waves read the image behind this barrier)
  }
  g8_barrier();     // (the tile waves' last barrier: their epilogue images go to the buffers, not to this wave's stages)
  // ---- what is left of the stream: free-running, one step ahead ----
  {
    step_mfma();       // the step read at the end of the last K step
    int sg = nk & 1;   // (stage sg ^ 1 holds the step issued in the last K step, stage sg is free)
    int issued = issue(sg);
    h8_wait(issued);
    step_reads(sg ^ 1);
    step_mfma();
    while (issued) {
      const int nxt = issue(sg ^ 1);
      h8_wait(nxt);
      step_reads(sg);
      step_mfma();
      sg ^= 1;
      issued = nxt;
    }
  }
}

// The tile with four helper waves (768 threads, three waves per SIMD): riders of the launch and / or T = A Ut^T by the helpers
template <class G, int EPI, int MODE, bool COLSUM>
__global__ __launch_bounds__(768, 3) void gemm8h_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const TsProblem t0,
                                                        const TsProblem t1, const int ldg, const int Mts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w >= 8) {
    g8_helper<G, MODE, COLSUM>(p, tiles_n, nwg, blockIdx.x, t0, t1, ldg, Mts, smem);
    return;
  }
  static_assert((G::NPA0 % 8) != 0 || (G::NPA1 % 8) != 0, "the 160-row tile: two wave classes");
  if (w < 4) g8_tile<G, EPI, 0, MODE>(p, tiles_n, nwg, blockIdx.x, smem);
  else g8_tile<G, EPI, 1, MODE>(p, tiles_n, nwg, blockIdx.x, smem);
}

TsProblem g8_problem(const cara_g8_product& q) {
  TsProblem t;
  t.X = static_cast<const bf16*>(q.X); t.Gt = static_cast<const bf16*>(q.Gt); t.slabs = q.slabs; t.cs_slabs = q.cs_slabs;
  t.ldx = q.ldx; t.K1 = q.K1; t.nchunks = q.nchunks; t.nblk = q.nblk;
  return t;
}

template <class G, int EPI, int MODE>
int g8_launch(const cara_gemm_args* a, hipStream_t st, const cara_g8_riders* ts, const bool helpers) {
  const int tiles_n = (a->N + 255) / 256, tiles_m = (a->M + G::MT - 1) / G::MT;
  const int nwg = tiles_m * tiles_n;
  if constexpr (G::MT == 160 && (MODE == 1 || MODE == 3)) {
    // helper waves: the riders' streams and, MODE 3, T = A Ut^T
    if (helpers && !a->er_Tt) {
      if (ts && !(EPI == CARA_EPI_BF16 || EPI == CARA_EPI_DGELU)) return -1;
      TsProblem t0 = {}, t1 = {};
      if (ts) { t0 = g8_problem(ts->a); t1 = g8_problem(ts->b); }
      const bool cs = ts && ts->any_cs;
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8h_kernel<G, EPI, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H8<G>::LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8h_kernel<G, EPI, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H8<G>::LDS) != hipSuccess)
          return CARA_E_LAUNCH;
        attr = true;
      }
      if (cs) hipLaunchKernelGGL((gemm8h_kernel<G, EPI, MODE, true>), dim3(nwg), dim3(768), H8<G>::LDS, st, *a, tiles_n, nwg, t0, t1, ts ? ts->ldg : 0, ts ? ts->M : 0);
      else hipLaunchKernelGGL((gemm8h_kernel<G, EPI, MODE, false>), dim3(nwg), dim3(768), H8<G>::LDS, st, *a, tiles_n, nwg, t0, t1, ts ? ts->ldg : 0, ts ? ts->M : 0);
      CARA_CHECK_LAUNCH();
      return CARA_OK;
    }
  }
  if constexpr (EPI == CARA_EPI_BF16 && (MODE == 1 || MODE == 2) && G::MT == 160) {
    if (a->er_Tt) {   // dVs (+ dc) of the GEMM's own linear out of its A sub-buffers (g8_tile<.., DV>); riders: a pending dU at most
      constexpr int TSB2 = 2 * TsRing<1, 3>::BLOCK_BYTES;
      constexpr int LDS_TS = G::LDS > TSB2 ? G::LDS : TSB2;
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<G, EPI, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_ts_kernel<G, EPI, MODE, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TS) != hipSuccess)
          return CARA_E_LAUNCH;
        attr = true;
      }
      if (ts) {
        if (ts->any_cs) return -1;
        const TsProblem t0 = g8_problem(ts->a), t1 = g8_problem(ts->b);
        const int nts = (t0.nblk + t1.nblk + 1) / 2;
        hipLaunchKernelGGL((gemm8_ts_kernel<G, EPI, MODE, false, true>), dim3(nwg + nts), dim3(512), LDS_TS, st, *a, tiles_n, nwg, t0, t1, ts->ldg, ts->M);
      } else {
        hipLaunchKernelGGL((gemm8_kernel<G, EPI, MODE, true>), dim3(nwg), dim3(512), G::LDS, st, *a, tiles_n, nwg, 0);
      }
      CARA_CHECK_LAUNCH();
      return CARA_OK;
    }
  }
  if (a->er_Tt) return -1;
  if (ts) {
    if constexpr ((EPI == CARA_EPI_BF16 || EPI == CARA_EPI_DGELU) && MODE != 3) {   // (the dX products; riders behind the tiles)
      constexpr int TSB2 = 2 * TsRing<1, 3>::BLOCK_BYTES;
      constexpr int LDS_TS = G::LDS > TSB2 ? G::LDS : TSB2;
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_ts_kernel<G, EPI, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_ts_kernel<G, EPI, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TS) != hipSuccess)
          return CARA_E_LAUNCH;
        attr = true;
      }
      const TsProblem t0 = g8_problem(ts->a), t1 = g8_problem(ts->b);
      const int nts = (t0.nblk + t1.nblk + 1) / 2;
      if (ts->any_cs) hipLaunchKernelGGL((gemm8_ts_kernel<G, EPI, MODE, true>), dim3(nwg + nts), dim3(512), LDS_TS, st, *a, tiles_n, nwg, t0, t1, ts->ldg, ts->M);
      else hipLaunchKernelGGL((gemm8_ts_kernel<G, EPI, MODE, false>), dim3(nwg + nts), dim3(512), LDS_TS, st, *a, tiles_n, nwg, t0, t1, ts->ldg, ts->M);
      CARA_CHECK_LAUNCH();
      return CARA_OK;
    } else {
      return -1;
    }
  }
  if constexpr (MODE == 3) {
    return -1;
  } else {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<G, EPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
        return CARA_E_LAUNCH;
      attr = true;
    }
    static const int stagger_env = [] { const char* e = getenv("CARA_GEMM8_STAGGER"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL((gemm8_kernel<G, EPI, MODE>), dim3(nwg), dim3(512), G::LDS, st, *a, tiles_n, nwg, nwg > 256 ? stagger_env : 0);
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
}

template <class G, int MODE>
int g8_launch_epi(const cara_gemm_args* a, hipStream_t st, const cara_g8_riders* ts, const bool helpers) {
  switch (a->epi) {
    case CARA_EPI_BF16: return g8_launch<G, CARA_EPI_BF16, MODE>(a, st, ts, helpers);
    case CARA_EPI_F32: return g8_launch<G, CARA_EPI_F32, MODE>(a, st, ts, helpers);
    case CARA_EPI_GELU: return g8_launch<G, CARA_EPI_GELU, MODE>(a, st, ts, helpers);
    case CARA_EPI_RESID: return g8_launch<G, CARA_EPI_RESID, MODE>(a, st, ts, helpers);
    case CARA_EPI_DGELU: return g8_launch<G, CARA_EPI_DGELU, MODE>(a, st, ts, helpers);
    default: return -1;
  }
}

}  // namespace

// Helper waves (gemm8h_kernel) are OFF by default (CARA_GEMM8_HELPERS=1 or cara_debug_set_gemm8_helpers(1) turns them on): in the
// step they lose to the plain tile -- T = A Ut^T by the helpers instead of the tile waves costs fc2 forward 5 us (twelve waves at
// every barrier), and the streamed riders of fc1 dX end at 92 us against 87 for the 128 x 128 x 32 kernel (profiles/r04_b_*).
// Without them the adapter inside is computed by the tile waves (MODE 2) and riding products run as workgroups behind the tiles.
static int g_helpers_override = -1;
bool cara_debug_setters_allowed();   // gemm.hip
extern "C" int cara_debug_set_gemm8_helpers(int on) {
  if (!cara_debug_setters_allowed()) return CARA_E_ARG;
  g_helpers_override = on;
  return CARA_OK;
}
static bool g8_helpers() {
  static const int env = [] { const char* e = getenv("CARA_GEMM8_HELPERS"); return e ? atoi(e) : 0; }();
  return (g_helpers_override >= 0 ? g_helpers_override : env) != 0;
}

bool cara_gemm8_helpers_on() { return g8_helpers(); }

int cara_gemm8_plan(const cara_gemm_args* a, int mt, int riders_nt) {
  // what the tile takes: row-major A and B (whole 128-byte lines per K step of 64); the K-extension at Rp = 32 with T given (A2) or
  // computed inside (Ut, rank <= 16); riding products of one r-tile (rank <= 16)
  if (a->a_panels || a->batch > 1 || a->B3 || a->K < 128 || (a->K % 64) || a->M < 1024 || (a->M % 16) || (a->N % 16)) return 0;
  if ((unsigned long long)a->M * a->lda * 2 >= 0x7fffff00ull || (unsigned long long)a->N * a->ldb * 2 >= 0x7fffff00ull) return 0;
  int mode = 0;
  if (a->Ut) {
    if (a->A2 || !a->B2 || a->Rp != 32 || !a->T_out || a->Ut_rank < 1 || a->Ut_rank > 16 || (a->Tt_out && (a->ldt < a->M || (a->ldt & 7)))) return 0;
    mode = 2;
  } else if (a->Rp) {
    if (a->Rp != 32 || !a->A2 || !a->B2) return 0;
    mode = 1;
  }
  const bool riders = riders_nt != 0;
  if (riders && (riders_nt != 1 || !(a->epi == CARA_EPI_BF16 || a->epi == CARA_EPI_DGELU))) return 0;
  if (a->er_Tt) {   // dVs out of the A sub-buffers: CARA_EPI_BF16, the 160-row tile, a K-extension (given or inside), 32-bit slab offsets
    if (a->epi != CARA_EPI_BF16 || mt != 160 || !mode || a->er_h || !a->er_slabs_v || a->er_ldg < a->M || (a->er_ldg & 7)) return 0;
    const unsigned long long chunks = (a->M + 159) / 160;
    if (chunks * (a->K / 64) * (64 * 16 * 4) >= 0x7fffff00ull) return 0;
    return 1;
  }
  if (mt == 256) return (mode || riders) ? 0 : 1;   // (the yardstick tile: plain products)
  if (mt != 160) return 0;
  if (g8_helpers() && (mode == 2 || (riders && mode == 1))) return 2;
  return 1;
}

int cara_gemm8_launch(const cara_gemm_args* a, hipStream_t st, int mt, const cara_g8_riders* ts) {
  const int plan = cara_gemm8_plan(a, mt, ts ? ts->nt : 0);
  if (!plan) return -1;
  if (mt == 256) return g8_launch_epi<G8<4, 4>, 0>(a, st, nullptr, false);
  const int mode = a->Ut ? 2 : (a->Rp ? 1 : 0);
  if (mode == 2) {
    // consumers read Tt in whole 32-row steps: keep columns [M, roundup32(M)) zero, as cara_skinny_xu does
    const int m32 = (a->M + 31) / 32 * 32;
    if (a->Tt_out && m32 > a->M && m32 <= a->ldt &&
        hipMemset2DAsync(static_cast<bf16*>(a->Tt_out) + a->M, (size_t)a->ldt * 2, 0, (size_t)(m32 - a->M) * 2, a->Rp, st) != hipSuccess)
      return CARA_E_LAUNCH;
    return plan == 2 ? g8_launch_epi<G8<3, 2>, 3>(a, st, ts, true) : g8_launch_epi<G8<3, 2>, 2>(a, st, ts, false);
  }
  return mode == 1 ? g8_launch_epi<G8<3, 2>, 1>(a, st, ts, plan == 2) : g8_launch_epi<G8<3, 2>, 0>(a, st, ts, false);
}
