// 256x256x64 bf16 MFMA GEMM tile for the large products of the adapted ViT block (gfx950).
//
// Why a second tile: on the 128x128 kernel of gemm.hip rocprofv3 shows each wave MFMA-busy for
// only ~20 % of its cycles at M = 12608 -- a 128^2 tile moves 32 KiB per 2.1 MFLOP, too little
// work per byte to cover the L2/HBM latency with one K-step of prefetch.  A 256^2 tile moves
// 64 KiB per 8.4 MFLOP (2x the FLOP per byte, 4x the MFMA work per barrier).
//
// Geometry: 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the output =
// 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 VGPRs), computed per K-tile as four
// 64 x 32 quadrants in the order (0,0) (0,1) (1,1) (1,0) so that only one operand's fragments
// change between quadrants.  LDS = 2 K-tile buffers x (A 256x64 + B 256x64) bf16 = 128 KiB
// (one workgroup per CU, 2 waves per SIMD).  Tiles are staged by 16-byte global_load_lds one
// whole K-tile ahead (issued before the MFMAs of the current K-tile), with the same
// source-side XOR swizzle as gemm.hip (conflict-free ds_read_b128 fragments).
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OP_BYTES = BM * BK * 2;      // 32 KiB per operand per K-tile
constexpr int BUF_BYTES = 2 * OP_BYTES;    // A + B
constexpr int LDS_BYTES = 2 * BUF_BYTES;   // 128 KiB

__device__ __forceinline__ int swz_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// rows r0..r0+255 (clamped), columns k0..k0+63 -> swizzled [256][64] image; 32 one-KiB pieces
// (8 rows each), wave w issues pieces 4w..4w+3
__device__ __forceinline__ void stage_op(const bf16* __restrict__ P, int ld, int r0, int rmax, int k0,
                                         char* img, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = wave * 4 + t;
    const int r = q * 8 + (lane >> 3);
    const int cg = (lane & 7) ^ ((r >> 1) & 7);
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;
    glds16(P + (size_t)gr * ld + k0 + cg * 8, img + q * 1024);
  }
}

__device__ __forceinline__ void stage_ext(const bf16* __restrict__ P, int Rp, int r0, int rmax, char* img, int tid) {
  const int cpr = Rp >> 3;
  for (int idx = tid; idx < 256 * cpr; idx += 512) {
    const int r = idx / cpr, c = idx - r * cpr;
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;
    *reinterpret_cast<uint4*>(img + swz_off(r, c)) = *reinterpret_cast<const uint4*>(P + (size_t)gr * Rp + c * 8);
  }
}

struct Frags {
  bf16x8 a[2][4];  // [kk][m-tile of the current 64-row half]
  bf16x8 b[2][2];  // [kk][n-tile of the current 32-col half]
};

template <int KSUB>
__device__ __forceinline__ void load_a(Frags& f, const char* sA, int row0, int fr, int fq) {
#pragma unroll
  for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[kk][i] = *reinterpret_cast<const bf16x8*>(sA + swz_off(row0 + i * 16 + fr, kk * 4 + fq));
}
template <int KSUB>
__device__ __forceinline__ void load_b(Frags& f, const char* sB, int row0, int fr, int fq) {
#pragma unroll
  for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
    for (int j = 0; j < 2; ++j) f.b[kk][j] = *reinterpret_cast<const bf16x8*>(sB + swz_off(row0 + j * 16 + fr, kk * 4 + fq));
}
template <int KSUB>
__device__ __forceinline__ void mma_quadrant(f32x4 (&acc)[8][4], const Frags& f, int qm, int qn) {
#pragma unroll
  for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[qm * 4 + i][qn * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[kk][i], f.b[kk][j], acc[qm * 4 + i][qn * 2 + j], 0, 0, 0);
}

// one K-tile (KSUB 32-wide sub-steps) of the wave's 128x64 output
template <int KSUB>
__device__ __forceinline__ void mma_ktile(f32x4 (&acc)[8][4], const char* sA, const char* sB, int wm, int wn, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const int ar = wm * 128, br = wn * 64;
  Frags f;
  load_a<KSUB>(f, sA, ar, fr, fq);
  load_b<KSUB>(f, sB, br, fr, fq);
  mma_quadrant<KSUB>(acc, f, 0, 0);
  load_b<KSUB>(f, sB, br + 32, fr, fq);
  mma_quadrant<KSUB>(acc, f, 0, 1);
  load_a<KSUB>(f, sA, ar + 64, fr, fq);
  mma_quadrant<KSUB>(acc, f, 1, 1);
  load_b<KSUB>(f, sB, br, fr, fq);
  mma_quadrant<KSUB>(acc, f, 1, 0);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const cara_gemm_args p, const int tiles_n, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* __restrict__ A = static_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = static_cast<const bf16*>(p.B);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_op(A, p.lda, m0, p.M - 1, 0, smem, wave, lane);
  stage_op(B, p.ldb, n0, p.N - 1, 0, smem + OP_BYTES, wave, lane);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // K-tile kt has landed (own DMA: vmcnt, other waves': barrier); every wave has issued the MFMAs
    // that consumed its fragment reads of the other buffer, so that buffer may be refilled.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* sA = smem + cur * BUF_BYTES;
    if (kt + 1 < nk) {
      char* nA = smem + (cur ^ 1) * BUF_BYTES;
      stage_op(A, p.lda, m0, p.M - 1, (kt + 1) * BK, nA, wave, lane);
      stage_op(B, p.ldb, n0, p.N - 1, (kt + 1) * BK, nA + OP_BYTES, wave, lane);
    }
    mma_ktile<2>(acc, sA, sA + OP_BYTES, wm, wn, lane);
    cur ^= 1;
  }
  if (p.Rp > 0) {
    __syncthreads();
    stage_ext(static_cast<const bf16*>(p.A2), p.Rp, m0, p.M - 1, smem, tid);
    stage_ext(static_cast<const bf16*>(p.B2), p.Rp, n0, p.N - 1, smem + OP_BYTES, tid);
    __syncthreads();
    if (p.Rp == 64) mma_ktile<2>(acc, smem, smem + OP_BYTES, wm, wn, lane);
    else mma_ktile<1>(acc, smem, smem + OP_BYTES, wm, wn, lane);
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
  hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(nwg), dim3(512), LDS_BYTES, st, *a, tiles_n, nwg);
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
