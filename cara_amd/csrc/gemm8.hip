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
//  * M = 12608 rows: 79 tiles of 160 rows x 3 column tiles = 237 workgroups = 92.6 % of the CUs in ONE round.
//
// Sequence numbers of the sub-buffers (what the counted waits are derived from): K step t holds seq 4t .. 4t + 3 =
// A0, B0, B1, A1; phase phi = 4t + (p - 1) issues seq phi + 6 and then waits until seq <= phi + 2 has landed (read
// one phase later), i.e. allows seq phi + 3 .. phi + 6 to stay in flight -- always one sub-buffer of each kind.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "gemm8.h"

namespace {

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
  static constexpr int LDS = 2 * BUF;
  static constexpr int NPA0 = A0_ROWS / 8, NPA1 = A1_ROWS / 8;   // one-KiB pieces (8 rows x 128 B)
  static_assert(NPA0 % 8 == 0 || NPA0 % 8 == 4, "piece counts: two wave classes at most");
  static_assert(NPA1 % 8 == 0 || NPA1 % 8 == 4, "piece counts: two wave classes at most");
  static_assert(A0_ROWS * 128 >= MT * 64, "the extension's T rows (64 B each) go to the A0 region");
};
// LDS-DMA instructions per wave and sub-buffer: waves 0..3 (CLS 0) issue one more than waves 4..7 where the pieces
// do not divide by 8
template <class G, int CLS>
struct G8Cnt {
  static constexpr int A0 = G::NPA0 / 8 + ((G::NPA0 % 8) && CLS == 0 ? 1 : 0);
  static constexpr int A1 = G::NPA1 / 8 + ((G::NPA1 % 8) && CLS == 0 ? 1 : 0);
  static constexpr int B = 2;
  static constexpr int ALL = A0 + A1 + 2 * B;
};

template <class G, int EPI, int CLS, bool EXT>
__device__ __forceinline__ void g8_tile(const cara_gemm_args& p, const int tiles_n, const int nwg, const int block, char* smem) {
  using C = G8Cnt<G, CLS>;
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
  const __amdgpu_buffer_rsrc_t rsEA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(EXT ? p.A2 : p.A), 0, (int)((unsigned)p.M * 64u), 0x00020000);
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
#define G8_ST_B1(KT, BUFX) G8_DMA(rsB, ((KT) < nk ? vB : vD), sB1[t_] + (KT) * 128, C::B, G::B1_OFF, BUFX)
#define G8_ST_A0(KT, BUFX)                                                                                                  \
  do {                                                                                                                     \
    const bool x_ = EXT && (KT) == nk;                                                                                     \
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
#define G8_MMA(I0, I1, J0, J1)                                                                                   \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                                          \
        _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_)                                                   \
            _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_)                                               \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i_][kh_], b[j_][kh_], acc[i_][j_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)
  // one phase behind its reads and DMA issue: counted wait (one sub-buffer of each kind stays in flight), barrier, this
  // quadrant's MFMAs, barrier
#define G8_SYNC_MMA(I0, I1, J0, J1) \
  do {                              \
    g8_vmcnt<C::ALL>();             \
    g8_barrier();                   \
    g8_lgkm0();                     \
    G8_MMA(I0, I1, J0, J1);         \
    g8_barrier();                   \
  } while (0)
  // K step T in buffer CUR
#define G8_TILE(T, CUR)                                \
  do {                                                 \
    G8_RD_A0(CUR) G8_RD_B0(CUR)                        \
    G8_ST_B1((T) + 1, (CUR) ^ 1);                      \
    G8_SYNC_MMA(0, RT0, 0, 2);                         \
    G8_RD_B1(CUR)                                      \
    G8_ST_A1((T) + 1, (CUR) ^ 1);                      \
    G8_SYNC_MMA(0, RT0, 2, 4);                         \
    G8_RD_A1(CUR)                                      \
    G8_ST_A0((T) + 2, CUR);                            \
    G8_SYNC_MMA(RT0, RT, 2, 4);                        \
    G8_ST_B0((T) + 2, CUR);                            \
    G8_SYNC_MMA(RT0, RT, 0, 2);                        \
  } while (0)

  const int nk = p.K >> 6;   // >= 2 (checked at dispatch)
  // prologue: K step 0 whole, A0 and B0 of K step 1
  G8_ST_A0(0, 0);
  G8_ST_B0(0, 0);
  G8_ST_B1(0, 0);
  G8_ST_A1(0, 0);
  G8_ST_A0(1, 1);
  G8_ST_B0(1, 1);
  g8_vmcnt<C::ALL>();
  g8_barrier();
  if (wr == 1) g8_barrier();   // the second wave row runs one barrier behind the first
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    G8_TILE(t, 0);
    G8_TILE(t + 1, 1);
  }
  if (t < nk) G8_TILE(t, 0);
  if (wr == 0) g8_barrier();
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
#undef G8_TILE
}

template <class G, int EPI, bool EXT>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const cara_gemm_args p, const int tiles_n, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool TWO_CLASSES = (G::NPA0 % 8) != 0 || (G::NPA1 % 8) != 0;
  if constexpr (TWO_CLASSES) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) g8_tile<G, EPI, 0, EXT>(p, tiles_n, nwg, blockIdx.x, smem);
    else g8_tile<G, EPI, 1, EXT>(p, tiles_n, nwg, blockIdx.x, smem);
  } else {
    g8_tile<G, EPI, 0, EXT>(p, tiles_n, nwg, blockIdx.x, smem);
  }
}

template <class G, int EPI, bool EXT>
int g8_launch(const cara_gemm_args* a, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<G, EPI, EXT>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
      return CARA_E_LAUNCH;
    attr = true;
  }
  const int tiles_n = (a->N + 255) / 256, tiles_m = (a->M + G::MT - 1) / G::MT;
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL((gemm8_kernel<G, EPI, EXT>), dim3(nwg), dim3(512), G::LDS, st, *a, tiles_n, nwg);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

template <class G, bool EXT>
int g8_launch_epi(const cara_gemm_args* a, hipStream_t st) {
  switch (a->epi) {
    case CARA_EPI_BF16: return g8_launch<G, CARA_EPI_BF16, EXT>(a, st);
    case CARA_EPI_F32: return g8_launch<G, CARA_EPI_F32, EXT>(a, st);
    case CARA_EPI_GELU: return g8_launch<G, CARA_EPI_GELU, EXT>(a, st);
    case CARA_EPI_RESID: return g8_launch<G, CARA_EPI_RESID, EXT>(a, st);
    case CARA_EPI_DGELU: return g8_launch<G, CARA_EPI_DGELU, EXT>(a, st);
    default: return -1;
  }
}

}  // namespace

int cara_gemm8_launch(const cara_gemm_args* a, hipStream_t st, int mt) {
  // what the tile takes: row-major A and B (whole 128-byte lines per K step of 64), the plain K-extension at Rp = 32
  if (a->a_panels || a->batch > 1 || a->B3 || a->Ut || a->K < 128 || (a->K % 64) || a->M < 1024 || (a->M % 16) || (a->N % 16)) return -1;
  if ((unsigned long long)a->M * a->lda * 2 >= 0x7fffff00ull || (unsigned long long)a->N * a->ldb * 2 >= 0x7fffff00ull) return -1;
  if (!(a->Rp == 0 || (a->Rp == 32 && a->A2 && a->B2))) return -1;
  if (mt == 256) {
    if (a->Rp) return -1;   // (the yardstick tile: plain products)
    return g8_launch_epi<G8<4, 4>, false>(a, st);
  }
  if (mt == 160) return a->Rp ? g8_launch_epi<G8<3, 2>, true>(a, st) : g8_launch_epi<G8<3, 2>, false>(a, st);
  return -1;
}
