// 256x256 bf16 MFMA GEMM tile with a deep LDS-DMA ring for the large products of the adapted ViT
// block (gfx950).
//
// Why: rocprofv3 on the 128x128x64 double-buffered kernel of gemm.hip shows each wave MFMA-busy
// for ~20 % of its cycles and parked ~37 % at the per-K-step wait: one K-step of prefetch does not
// cover the load latency seen under load (L2 hit ~80 % of reads, the rest comes from the Infinity
// Cache / HBM).  This kernel keeps THREE K-tiles in flight per workgroup:
//
//   tile 256x256, BK = 32, ring of 4 LDS slots x (A 256x32 + B 256x32) bf16 = 4 x 32 KiB = 128 KiB
//   (one workgroup per CU), 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 =
//   8 x 4 accumulators of v_mfma_f32_16x16x32_bf16; per K-tile a wave issues 4 LDS-DMA pieces,
//   reads 12 fragments (ds_read_b128, conflict-free XOR swizzle) and issues 32 MFMAs.
//
// Loop invariant at iteration kt: the fragments of tile kt are already in registers (read during
// iteration kt-1); tile kt+1 has landed (own pieces: counted vmcnt(8) leaves kt+2, kt+3 in flight;
// other waves': the barrier); every wave has finished reading slot kt%4 (it read it before this
// barrier), so tile kt+4 is issued into it; then the fragment reads of tile kt+1 overlap the 32
// MFMAs of tile kt (two register sets, loop unrolled by two so that they are statically named).
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NSLOT = 4;
constexpr int OP_BYTES = BM * BK * 2;      // 16 KiB per operand per K-tile
constexpr int SLOT_BYTES = 2 * OP_BYTES;   // A + B
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES;   // 128 KiB
constexpr int PIECES = 4;                  // LDS-DMA instructions per wave per K-tile

// [256 rows][4 chunks of 16 B]; chunk' = chunk ^ (row & 8 ? 3 : 0): the 16 lanes that a ds_read_b128
// services together (4 row classes mod 4 x 4 chunks) then hit 64 distinct banks
__device__ __forceinline__ int swz32(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) * 3)) << 4); }
__device__ __forceinline__ int swz64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// one K-tile of both operands: 2 x 16 pieces of 1 KiB (16 rows x 64 B); wave w issues A pieces
// 2w, 2w+1 and B pieces 2w, 2w+1
__device__ __forceinline__ void stage_ktile(const bf16* __restrict__ A, int lda, int m0, int mmax,
                                            const bf16* __restrict__ B, int ldb, int n0, int nmax, int k0,
                                            char* slot, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int q = wave * 2 + t;
    const int r = q * 16 + (lane >> 2);
    const int cg = (lane & 3) ^ (((r >> 3) & 1) * 3);
    int ga = m0 + r, gb = n0 + r;
    ga = ga < mmax ? ga : mmax;
    gb = gb < nmax ? gb : nmax;
    glds16(A + (size_t)ga * lda + k0 + cg * 8, slot + q * 1024);
    glds16(B + (size_t)gb * ldb + k0 + cg * 8, slot + OP_BYTES + q * 1024);
  }
}

// the K-extension operands ([rows, Rp], Rp = 32 or 64) go through registers into a [256][64]
// image with the 128-byte-row swizzle
__device__ __forceinline__ void stage_ext(const bf16* __restrict__ P, int Rp, int r0, int rmax, char* img, int tid) {
  const int cpr = Rp >> 3;
  for (int idx = tid; idx < 256 * cpr; idx += 512) {
    const int r = idx / cpr, c = idx - r * cpr;
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;
    *reinterpret_cast<uint4*>(img + swz64(r, c)) = *reinterpret_cast<const uint4*>(P + (size_t)gr * Rp + c * 8);
  }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt");
}

struct Frags {
  bf16x8 a[8], b[4];
};

template <bool EXT>
__device__ __forceinline__ void load_frags(Frags& f, const char* sA, const char* sB, int wm, int wn, int lane, int kk) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wm * 128 + i * 16 + fr;
    f.a[i] = *reinterpret_cast<const bf16x8*>(sA + (EXT ? swz64(row, kk * 4 + fq) : swz32(row, fq)));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wn * 64 + j * 16 + fr;
    f.b[j] = *reinterpret_cast<const bf16x8*>(sB + (EXT ? swz64(row, kk * 4 + fq) : swz32(row, fq)));
  }
}

__device__ __forceinline__ void mma_frags(f32x4 (&acc)[8][4], const Frags& f) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
}

template <bool EXT>
__device__ __forceinline__ void mma_kstep(f32x4 (&acc)[8][4], const char* sA, const char* sB, int wm, int wn, int lane, int kk) {
  Frags f;
  load_frags<EXT>(f, sA, sB, wm, wn, lane, kk);
  mma_frags(acc, f);
}

// one pipelined iteration: fragments of tile kt are in `cur`; make tile kt+1 visible, refill the
// slot tile kt was read from with tile kt+4, read tile kt+1's fragments into `nxt` WHILE the 32
// MFMAs of tile kt execute (the two are independent, the compiler interleaves them).
__device__ __forceinline__ void pipe_step(f32x4 (&acc)[8][4], const Frags& cur, Frags& nxt, int kt, int nk,
                                          const bf16* __restrict__ A, int lda, int m0, int mmax,
                                          const bf16* __restrict__ B, int ldb, int n0, int nmax,
                                          char* smem, int wm, int wn, int wave, int lane, const int ablate) {
  const int after = nk - 2 - kt;   // tiles issued after kt+1 that may stay in flight: min(2, after)
  if (after >= 2) wait_vm<2 * PIECES>();
  else if (after == 1) wait_vm<PIECES>();
  else wait_vm<0>();
  if (!(ablate & 8)) __syncthreads();
  if (kt + 4 < nk && !(ablate & 1))
    stage_ktile(A, lda, m0, mmax, B, ldb, n0, nmax, (kt + 4) * BK, smem + (kt & 3) * SLOT_BYTES, wave, lane);
  if (kt + 1 < nk && !(ablate & 2)) {
    const char* sA = smem + ((kt + 1) & 3) * SLOT_BYTES;
    load_frags<false>(nxt, sA, sA + OP_BYTES, wm, wn, lane, 0);
  }
  mma_frags(acc, cur);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const cara_gemm_args p, const int tiles_n, const int nwg,
                                                         const int ablate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* __restrict__ A = static_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = static_cast<const bf16*>(p.B);
  const int mmax = p.M - 1, nmax = p.N - 1;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;   // even, >= 2 (K % 64 == 0)
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nk) stage_ktile(A, p.lda, m0, mmax, B, p.ldb, n0, nmax, t * BK, smem + t * SLOT_BYTES, wave, lane);
  // tile 0 landed: up to 3 younger tiles may stay in flight
  if (nk >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else wait_vm<PIECES>();   // nk == 2: one younger tile
  __syncthreads();
  Frags fa, fb;
  load_frags<false>(fa, smem, smem + OP_BYTES, wm, wn, lane, 0);
  if (ablate & 2) load_frags<false>(fb, smem, smem + OP_BYTES, wm, wn, lane, 0);
  if (ablate & 16) return;  // timing-only: launch + prologue
  for (int kt = 0; kt < nk; kt += 2) {
    pipe_step(acc, fa, fb, kt, nk, A, p.lda, m0, mmax, B, p.ldb, n0, nmax, smem, wm, wn, wave, lane, ablate);
    pipe_step(acc, fb, fa, kt + 1, nk, A, p.lda, m0, mmax, B, p.ldb, n0, nmax, smem, wm, wn, wave, lane, ablate);
  }
  if (ablate & 4) {   // timing-only: no K-extension, no epilogue (one store keeps the accumulators live)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) static_cast<float*>(p.C)[0] = s;
    return;
  }
  if (p.Rp > 0) {
    __syncthreads();
    stage_ext(static_cast<const bf16*>(p.A2), p.Rp, m0, mmax, smem, tid);
    stage_ext(static_cast<const bf16*>(p.B2), p.Rp, n0, nmax, smem + 2 * OP_BYTES, tid);
    __syncthreads();
    for (int kk = 0; kk < (p.Rp >> 5); ++kk) mma_kstep<true>(acc, smem, smem + 2 * OP_BYTES, wm, wn, lane, kk);
  }

  // ---- epilogue: two 64-row halves through a wave-private 64x64 fp32 LDS image ----
  __syncthreads();
  float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 64);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[half * 4 + i][j][r];
    epilogue_64x64<EPI>(p, stg, m0 + wm * 128 + half * 64, n0 + wn * 64, lane);
  }
}

template <int EPI>
int launch256(const cara_gemm_args* a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  // CARA_GEMM_ABLATE (diagnostic, results become wrong): 1 = no in-loop DMA, 2 = no in-loop fragment
  // reads, 4 = stop before the K-extension/epilogue
  static int ablate = -1;
  if (ablate < 0) {
    const char* e = getenv("CARA_GEMM_ABLATE");
    ablate = e ? atoi(e) : 0;
  }
  hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(nwg), dim3(512), LDS_BYTES, st, *a, tiles_n, nwg, ablate);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

}  // namespace

// internal entry used by cara_gemm_bf16 (arguments already validated there)
int cara_gemm256_dispatch(const cara_gemm_args* a, hipStream_t st) {
  switch (a->epi) {
    case CARA_EPI_BF16: return launch256<CARA_EPI_BF16>(a, st);
    case CARA_EPI_F32: return launch256<CARA_EPI_F32>(a, st);
    case CARA_EPI_GELU: return launch256<CARA_EPI_GELU>(a, st);
    case CARA_EPI_RESID: return launch256<CARA_EPI_RESID>(a, st);
    case CARA_EPI_DGELU: return launch256<CARA_EPI_DGELU>(a, st);
    default: return CARA_E_ARG;
  }
}
