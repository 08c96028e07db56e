// HBM-bound rank-R adapter contractions (gfx950 / MI355X only).
//
//  cara_skinny_xu  : T[M,Rp]   = X[M,K] . U[K,Rp]        forward T = X U ; backward G' = dY Vs
//  cara_tskinny_xtg: D[K1,Rp]  = X[M,K1]^T . G[M,Rp]      dU = X^T G'  and  dVs = dY^T T  (A.4)
//
// Together with the K-extension of gemm.hip these replace the reference's materialise-dW +
// second dense GEMM + dense d(dW) (/root/reference/src/cara/cara.py:26-35,51-57,76-81,88-92 and
// their autograd) by products whose cost is one streaming read of X (algorithmic bytes:
// M*K*2 in, tiny out).  Both use MFMA so the VALU never bounds them.
#include "common.h"
#include "tskinny_body.h"
#include "gemm8.h"

namespace {

// ------------------------------------------------------------------------------------------
// T = X . Ut^T.  A workgroup owns 64 rows (4 groups of 16); wave w owns the K-slice
// [192 w, 192 w + 192) and keeps its Ut fragments (6 k-steps x NT) in registers for all four row
// groups, so X is the only stream (v1 re-read Ut from L2 for every 16 rows: two thirds of its
// load instructions).  The A fragments of the next row group are issued before the MFMAs of the
// current one; per group the K-slices are summed through LDS and written as T (row-major bf16)
// and T^T (the MFMA C layout holds 4 consecutive rows per lane: 8-byte stores).
// Waves per workgroup = K / 192 (4 / 12 / 16 for K = 768 / 2304 / 3072); other K fall back to v1.
// ------------------------------------------------------------------------------------------
constexpr int XU_KSLICE = 192, XU_GROUPS = 4;

// Rp = 64 (rank > 32): the Ut fragments of 64 columns would be 96 VGPRs per wave, over the budget of a 1024-thread
// workgroup, so the columns are cut into halves of 32 handled by TWO workgroups of the same rows (blockIdx.y = half:
// rows [32 y, 32 y + 32) of Ut, columns [32 y, ...) of T whose row stride is ldT = Rp); they run side by side, so the
// second read of a row block is served by the caches rather than by HBM.
template <int NT>
__global__ __launch_bounds__(1024) void skinny_xu_sliced_kernel(const bf16* __restrict__ X, int ldx,
                                                                const bf16* __restrict__ Ut,
                                                                bf16* __restrict__ T, bf16* __restrict__ Tt,
                                                                int ldt, int M, int K, const int ldT, const int groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);   // [nwaves][NT][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int KS = XU_KSLICE / 32;
  const int Rp = ldT;
  Ut += (size_t)blockIdx.y * (NT * 16) * K;
  T += blockIdx.y * (NT * 16);
  if (Tt) Tt += (size_t)blockIdx.y * (NT * 16) * ldt;
  const int k0 = wave * XU_KSLICE + fq * 8;
  bf16x8 u[KS][NT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NT; ++j) u[ks][j] = *reinterpret_cast<const bf16x8*>(Ut + (size_t)(j * 16 + fr) * K + k0 + ks * 32);
  const int mblk = blockIdx.x * (16 * groups);   // groups = XU_GROUPS row groups of 16 per workgroup; 1 for few-row products
  bf16x8 a[KS], an[KS];
  // ldx < 0: X is K-panel-major, [K/32][-ldx rows][32] (cara_gemm_args::c_panels): element (r, k0 + 32 ks) sits at
  // ((wave * KS + ks) * rows + r) * 32 + fq * 8 -- a wave's load is one contiguous KiB
  const size_t prow = ldx < 0 ? (size_t)(-ldx) : 0;
  auto xptr = [&](int r, int ks) {
    return prow ? X + ((size_t)(wave * KS + ks) * prow + r) * 32 + fq * 8 : X + (size_t)r * ldx + k0 + ks * 32;
  };
  {
    int r = mblk + fr;
    r = r < M ? r : M - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(xptr(r, ks));
  }
#pragma unroll
  for (int g = 0; g < XU_GROUPS; ++g) {
    if (g >= groups) break;
    const int m0 = mblk + g * 16;
    if (g + 1 < groups) {
      int r = m0 + 16 + fr;
      r = r < M ? r : M - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) an[ks] = *reinterpret_cast<const bf16x8*>(xptr(r, ks));
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], u[ks][j], acc[j], 0, 0, 0);
    if (g > 0) __syncthreads();   // the previous group's sums have been read
#pragma unroll
    for (int j = 0; j < NT; ++j) red[(wave * NT + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (m0 < M) {
      // (rank <= 16 at Rp = 32: NT = 1 computes the first 16 columns only -- the rows of Ut beyond the rank are zero -- and the
      // column tiles it skips are written as the zeros they are, by the waves behind the summing ones)
      for (int nt = NT + wave; nt < Rp / 16 && gridDim.y == 1; nt += nwaves) {
        const int n = nt * 16 + fr;
        const int mb = m0 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mb + r >= M) continue;
          T[(size_t)(mb + r) * Rp + n] = (bf16)0.f;
          if (Tt) Tt[(size_t)n * ldt + mb + r] = (bf16)0.f;
        }
      }
      for (int nt = wave; nt < NT; nt += nwaves) {
        f32x4 s = red[nt * 64 + lane];
        for (int w = 1; w < nwaves; ++w) {
          const f32x4 v = red[(w * NT + nt) * 64 + lane];
          s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        const int n = nt * 16 + fr;
        const int mb = m0 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mb + r < M) T[(size_t)(mb + r) * Rp + n] = (bf16)s[r];
        if (Tt) {
          if (mb + 3 < M) {
            bf16x4 pk = {(bf16)s[0], (bf16)s[1], (bf16)s[2], (bf16)s[3]};
            *reinterpret_cast<bf16x4*>(Tt + (size_t)n * ldt + mb) = pk;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (mb + r < M) Tt[(size_t)n * ldt + mb + r] = (bf16)s[r];
          }
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
  }
}

// v1 (any K % 32 == 0): block = 16 rows, the 4 waves take k-steps w, w+4, ... and reduce through LDS.
template <int NT>  // Rp / 16
__global__ __launch_bounds__(256) void skinny_xu_kernel(const bf16* __restrict__ X, int ldx,
                                                        const bf16* __restrict__ Ut,
                                                        bf16* __restrict__ T, bf16* __restrict__ Tt,
                                                        int ldt, int M, int K) {
  __shared__ f32x4 red[4][NT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  constexpr int Rp = NT * 16;
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int r0 = m0 + fr;
  r0 = r0 < M ? r0 : M - 1;
  // ldx < 0: X is K-panel-major, [K/32][-ldx rows][32]: K step ks of row r0 sits at (ks * rows + r0) * 32
  const bf16* xa = ldx < 0 ? X + (size_t)r0 * 32 + fq * 8 : X + (size_t)r0 * ldx + fq * 8;
  const size_t kstride = ldx < 0 ? (size_t)(-ldx) * 32 : 32;
  const bf16* ub = Ut + (size_t)fr * K + fq * 8;
  const int nks = K >> 5;
  // wave w takes k-steps w, w+4, ...: the four waves read adjacent 64-B pieces of each row
#pragma unroll 6
  for (int ks = wave; ks < nks; ks += 4) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(xa + ks * kstride);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(ub + (size_t)j * 16 * K + ks * 32);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) red[wave][j][lane] = acc[j];
  __syncthreads();
  for (int nt = wave; nt < NT; nt += 4) {
    f32x4 s = red[0][nt][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = red[w][nt][lane];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int n = nt * 16 + fr;
    const int mb = m0 + fq * 4;  // rows mb..mb+3, column n
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (mb + r < M) T[(size_t)(mb + r) * Rp + n] = (bf16)s[r];
    if (Tt) {
      if (mb + 3 < M) {
        bf16x4 pk = {(bf16)s[0], (bf16)s[1], (bf16)s[2], (bf16)s[3]};
        *reinterpret_cast<bf16x4*>(Tt + (size_t)n * ldt + mb) = pk;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mb + r < M) Tt[(size_t)n * ldt + mb + r] = (bf16)s[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// D[i, r] = sum_m X[m, i] * G[m, r]   with Gt[r, m] given.
// MFMA orientation: rows = r (A operand = Gt, m-contiguous), cols = i (B operand = X, taken
// from an LDS image of the [32 m][64 i] tile by ds_read_b64_tr_b16: k = m is the slow index of X).
// Grid = (K1/64 column blocks) x (m-chunks); inside a block the 4 waves take alternate 32-row
// steps, each with a PRIVATE 3-deep LDS ring filled by LDS-DMA (no block barrier in the loop,
// only counted vmcnt), and combine through LDS at the end.  Each block writes one fp32 slab
// [64, Rp]; a second tiny kernel sums the slabs in a fixed order (bitwise reproducible, no
// float atomics).
// ------------------------------------------------------------------------------------------
template <int NT, bool COLSUM, int NSTAGE = 3>
__global__ __launch_bounds__(256) void tskinny_kernel(const TsProblem p0, const TsProblem p1, int ldg, int M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tskinny_body<NT, COLSUM, NSTAGE>(p0, p1, ldg, M, blockIdx.x, smem);
}

// sum of n values `stride` floats apart, in a FIXED order that keeps four loads in flight: four interleaved partial
// sums, then (s0 + s1) + (s2 + s3)
__device__ __forceinline__ float strided_sum4(const float* __restrict__ p, const size_t stride, const int n) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= n; c += 4) {
    s0 += p[(size_t)c * stride];
    s1 += p[(size_t)(c + 1) * stride];
    s2 += p[(size_t)(c + 2) * stride];
    s3 += p[(size_t)(c + 3) * stride];
  }
  if (c < n) s0 += p[(size_t)c * stride];
  if (c + 1 < n) s1 += p[(size_t)(c + 1) * stride];
  if (c + 2 < n) s2 += p[(size_t)(c + 2) * stride];
  return (s0 + s1) + (s2 + s3);
}

// D[b][i][r] = sum_chunk slab[b][chunk*colblocks + i/64][i%64][r]   (fixed order); grid.y = batch
__global__ void tskinny_reduce_kernel(const char* __restrict__ slabs_base, size_t slab_stride,
                                      float* __restrict__ D, float* __restrict__ colsum, int K1, int Rp,
                                      int nchunks) {
  const int colblocks = K1 / TS_COLS;
  const int nblk = colblocks * nchunks;
  const float* slabs = reinterpret_cast<const float*>(slabs_base + (size_t)blockIdx.y * slab_stride);
  const float* cs_slabs = slabs + (size_t)nblk * TS_COLS * Rp;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = K1 * Rp;
  if (idx < total) {
    const int i = idx / Rp, r = idx - i * Rp;
    const int cb = i / TS_COLS, il = i - cb * TS_COLS;
    const float s = strided_sum4(slabs + ((size_t)cb * TS_COLS + il) * Rp + r, (size_t)colblocks * TS_COLS * Rp, nchunks);
    D[(size_t)blockIdx.y * total + idx] = s;
  }
  if (colsum && idx < K1) {
    const int cb = idx / TS_COLS, il = idx - cb * TS_COLS;
    const float s = strided_sum4(cs_slabs + (size_t)cb * TS_COLS + il, (size_t)colblocks * TS_COLS, nchunks);
    colsum[(size_t)blockIdx.y * K1 + idx] = s;
  }
}

// several reductions in ONE launch (grid.z = problem): the 8 + 6 slab sums at the end of a backward pass are 10 us
// each as separate launches, almost all of it launch latency
struct TsReduceTable {
  cara_ts_reduce p[CARA_TS_REDUCE_MAX];
  int nchunks[CARA_TS_REDUCE_MAX];
};
// (four columns per thread, 16-byte loads: a wave reads 512 contiguous bytes of a slab per instruction instead of 128; the sums
// are the same four interleaved partial sums per column, in the same order)
__device__ __forceinline__ f32x4 strided_sum4_v(const float* __restrict__ p, const size_t stride, const int n) {
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int c = 0;
  for (; c + 4 <= n; c += 4) {   // (unrolled twice -- eight loads in flight per thread -- the launch measured 38.7 us instead of 32.2: r05)
    s0 += *reinterpret_cast<const f32x4*>(p + (size_t)c * stride);
    s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(c + 1) * stride);
    s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(c + 2) * stride);
    s3 += *reinterpret_cast<const f32x4*>(p + (size_t)(c + 3) * stride);
  }
  if (c < n) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)c * stride);
  if (c + 1 < n) s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(c + 1) * stride);
  if (c + 2 < n) s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(c + 2) * stride);
  return (s0 + s1) + (s2 + s3);
}
__global__ void tskinny_reduce_many_kernel(const TsReduceTable t) {
  const cara_ts_reduce& q = t.p[blockIdx.z];
  if ((int)blockIdx.y >= q.batch) return;
  const int K1 = q.K1, Rp = q.Rp, nchunks = t.nchunks[blockIdx.z];
  const int Rc = q.Rc > 0 ? q.Rc : Rp;   // columns the slabs hold (16 when the products ran at rank <= 16): the rest of D is zero
  const int colblocks = K1 / TS_COLS;
  const int nblk = colblocks * nchunks;
  const float* slabs = reinterpret_cast<const float*>(static_cast<const char*>(q.slabs) + (size_t)blockIdx.y * q.slab_stride);
  const float* cs_slabs = slabs + (size_t)nblk * TS_COLS * Rp;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per four columns of a row of D
  const int quads = Rp >> 2, total = K1 * quads;
  if (idx < total) {
    const int i = idx / quads, r = (idx - i * quads) * 4;
    const int cb = i / TS_COLS, il = i - cb * TS_COLS;
    const f32x4 s = r < Rc ? strided_sum4_v(slabs + ((size_t)cb * TS_COLS + il) * Rc + r, (size_t)colblocks * TS_COLS * Rc, nchunks) : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(q.D + (size_t)blockIdx.y * K1 * Rp + (size_t)i * Rp + r) = s;
  }
  if (q.colsum && idx < K1) {
    const int cb = idx / TS_COLS, il = idx - cb * TS_COLS;
    const float s = strided_sum4(cs_slabs + (size_t)cb * TS_COLS + il, (size_t)colblocks * TS_COLS, nchunks);
    q.colsum[(size_t)blockIdx.y * K1 + idx] = s;
  }
}

}  // namespace

extern "C" int cara_tskinny_reduce_many(const cara_ts_reduce* probs, int n, void* stream) {
  if (!probs || n <= 0 || n > CARA_TS_REDUCE_MAX) return CARA_E_ARG;
  TsReduceTable t;
  int maxblocks = 0, maxbatch = 0;
  for (int i = 0; i < n; ++i) {
    const cara_ts_reduce& q = probs[i];
    if (!q.slabs || !q.D || q.batch <= 0 || q.M <= 0 || q.K1 <= 0 || (q.K1 % TS_COLS) || !(q.Rp == 32 || q.Rp == 64)) return CARA_E_ARG;
    if (!(q.Rc == 0 || q.Rc == q.Rp || (q.Rc == 16 && q.Rp == 32))) return CARA_E_ARG;
    if (((uintptr_t)q.slabs & 15) || ((uintptr_t)q.D & 15) || (q.slab_stride & 15)) return CARA_E_ARG;   // (16-byte accesses)
    t.p[i] = q;
    if (q.wave_slabs < 0 || (q.wave_slabs && !(q.Rc == 16 && q.Rp == 32))) return CARA_E_ARG;   // (only the one-r-tile products are written per wave / per row tile)
    // slab (chunk * 4 + wave) of a column block; >= 2: the epilogue riders of a GEMM left that many slabs per column block
    t.nchunks[i] = q.wave_slabs >= 2 ? q.wave_slabs : ts_chunks(q.M, q.K1) * (q.wave_slabs ? 4 : 1);
    const int blocks = (q.K1 * (q.Rp / 4) + 255) / 256;   // (>= K1 / 256: the column sums' threads are covered)
    maxblocks = blocks > maxblocks ? blocks : maxblocks;
    maxbatch = q.batch > maxbatch ? q.batch : maxbatch;
  }
  hipLaunchKernelGGL(tskinny_reduce_many_kernel, dim3(maxblocks, maxbatch, n), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_skinny_xu(const void* X, int ldx, const void* Ut, void* T, void* Tt, int ldt,
                              int M, int K, int Rp, void* stream) {
  return cara_skinny_xu_r(X, ldx, Ut, T, Tt, ldt, M, K, Rp, Rp, stream);
}

extern "C" int cara_skinny_xu_r(const void* X, int ldx, const void* Ut, void* T, void* Tt, int ldt,
                                int M, int K, int Rp, int rank, void* stream) {
  if (!X || !Ut || !T || M <= 0 || K <= 0 || (K & 31) || rank <= 0 || rank > Rp) return CARA_E_ARG;
  // ldx < 0: X is K-panel-major with -ldx >= M rows per panel (sliced kernel only)
  const bool xpanels = ldx < 0;
  if (xpanels ? -ldx < M : ((ldx & 7) || ldx < K)) return CARA_E_ARG;
  if (Tt && (ldt < M || (ldt & 7))) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // consumers (cara_tskinny_*) read Tt in whole 32-row steps: keep columns [M, roundup32(M)) zero even
  // when the buffer is shared between products of different M
  const int m32 = (M + 31) / 32 * 32;
  if (Tt && m32 > M && m32 <= ldt &&
      hipMemset2DAsync(static_cast<bf16*>(Tt) + M, (size_t)ldt * 2, 0, (size_t)(m32 - M) * 2, Rp, st) != hipSuccess)
    return CARA_E_LAUNCH;
  const bf16* x = (const bf16*)X;
  const bf16* u = (const bf16*)Ut;
  // (Rp = 64: two workgroups per row block, 32 columns each -- see the kernel)
  if ((Rp == 32 || Rp == 64) && K % XU_KSLICE == 0 && K / XU_KSLICE <= 16 && M >= 64) {
    const int nw = K / XU_KSLICE;
    // a few rows (the cls-row-only linears of the last block: M = batch): one row group per workgroup, or a single CU
    // walks all of them one after the other (20 us for 64 rows x 3072 columns; 6 us as four workgroups)
    const int groups = M <= 16 * XU_GROUPS * 8 ? 1 : XU_GROUPS;
    const dim3 g2((M + 16 * groups - 1) / (16 * groups), Rp / 32), b2(nw * 64);
    const size_t lds = (size_t)nw * 2 * 64 * sizeof(f32x4);
    if (Rp == 32 && rank <= 16)   // half of the padded rank is structurally zero: one column tile, half the Ut fragments
      hipLaunchKernelGGL(skinny_xu_sliced_kernel<1>, dim3(g2.x, 1), b2, lds, st, x, ldx, u, (bf16*)T, (bf16*)Tt, ldt, M, K, Rp, groups);
    else
      hipLaunchKernelGGL(skinny_xu_sliced_kernel<2>, g2, b2, lds, st, x, ldx, u, (bf16*)T, (bf16*)Tt, ldt, M, K, Rp, groups);
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  const dim3 grid((M + 15) / 16), block(256);
  if (Rp == 32)
    hipLaunchKernelGGL(skinny_xu_kernel<2>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)Ut, (bf16*)T, (bf16*)Tt, ldt, M, K);
  else if (Rp == 64)
    hipLaunchKernelGGL(skinny_xu_kernel<4>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)Ut, (bf16*)T, (bf16*)Tt, ldt, M, K);
  else
    return CARA_E_ARG;
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" size_t cara_tskinny_scratch_bytes(int M, int K1, int Rp) {
  if (M <= 0 || K1 <= 0 || (K1 % TS_COLS) || !(Rp == 32 || Rp == 64)) return 0;
  // One slab per tskinny BLOCK.  With helper waves on (CARA_GEMM8_HELPERS=1, off by default) a product that rides in a launch of the
  // 160 x 256 x 64 tile writes one slab per WAVE (cara_ts_reduce::wave_slabs): four times the blocks, the slabs spaced as 32-column
  // ones (gemm8.hip: the column sums sit behind 4 nblk slabs of TS_COLS x 32) -- four times the bytes; that form exists at Rp = 32,
  // rank <= 16 only.  Sizing every region 4x whatever the mode cost ~0.6 GB of workspace
  // at ViT-B batch 64 and ~1.3 GB at ViT-L batch 32 for a path that is off.  The switch is read when the workspace is
  // SIZED: flipping the debug setter between sizing and use is the caller's bug.
  const size_t blocks = (size_t)(K1 / TS_COLS) * ts_chunks(M, K1);
  const size_t one = blocks * TS_COLS * Rp * sizeof(float) + blocks * TS_COLS * sizeof(float);
  return (cara_gemm8_helpers_on() && Rp == 32) ? 4 * one : one;
}

namespace {
int ts_launch(const TsProblem& a, const TsProblem& b, int ldg, int M, int Rp, bool any_cs, hipStream_t st, bool half = false, bool small = false) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<4>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<4>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<2>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<2>::WAVE_BYTES);
    attr_set = true;
  }
  const dim3 grid(a.nblk + b.nblk), block(256);
  if (Rp == 32 && half && small) {   // the same with a two-stage ring per wave: 40 KiB of LDS, a block fits on a CU that holds a gemm8 tile
    const size_t lds = TsRing<1, 2>::BLOCK_BYTES;
    if (any_cs) hipLaunchKernelGGL((tskinny_kernel<1, true, 2>), grid, block, lds, st, a, b, ldg, M);
    else hipLaunchKernelGGL((tskinny_kernel<1, false, 2>), grid, block, lds, st, a, b, ldg, M);
  } else if (Rp == 32 && half) {   // rank <= 16: one r-tile (16 of the 32 columns), 16-wide slabs
    const size_t lds = TsRing<1>::BLOCK_BYTES;
    if (any_cs) hipLaunchKernelGGL((tskinny_kernel<1, true>), grid, block, lds, st, a, b, ldg, M);
    else hipLaunchKernelGGL((tskinny_kernel<1, false>), grid, block, lds, st, a, b, ldg, M);
  } else if (Rp == 32) {
    const size_t lds = 4 * TsRing<2>::WAVE_BYTES;
    if (any_cs) hipLaunchKernelGGL((tskinny_kernel<2, true>), grid, block, lds, st, a, b, ldg, M);
    else hipLaunchKernelGGL((tskinny_kernel<2, false>), grid, block, lds, st, a, b, ldg, M);
  } else {
    const size_t lds = 4 * TsRing<4>::WAVE_BYTES;
    if (any_cs) hipLaunchKernelGGL((tskinny_kernel<4, true>), grid, block, lds, st, a, b, ldg, M);
    else hipLaunchKernelGGL((tskinny_kernel<4, false>), grid, block, lds, st, a, b, ldg, M);
  }
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
}  // namespace

extern "C" int cara_tskinny_partial(const void* X, int ldx, const void* Gt, int ldg, void* slabs, int want_colsum,
                                    int M, int K1, int Rp, void* stream) {
  if (!ts_args_ok(X, ldx, Gt, ldg, slabs, M, K1, Rp)) return CARA_E_ARG;
  const TsProblem a = ts_problem(X, ldx, Gt, slabs, want_colsum, M, K1, Rp);
  TsProblem none = a;
  none.nblk = 0;
  return ts_launch(a, none, ldg, M, Rp, want_colsum != 0, static_cast<hipStream_t>(stream));
}

extern "C" int cara_tskinny_partial2(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                                     const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                                     int ldg, int M, int Rp, void* stream) {
  if (!ts_args_ok(Xa, ldxa, Gta, ldg, slabs_a, M, K1a, Rp) || !ts_args_ok(Xb, ldxb, Gtb, ldg, slabs_b, M, K1b, Rp)) return CARA_E_ARG;
  const TsProblem a = ts_problem(Xa, ldxa, Gta, slabs_a, 0, M, K1a, Rp);
  const TsProblem b = ts_problem(Xb, ldxb, Gtb, slabs_b, want_colsum_b, M, K1b, Rp);
  return ts_launch(a, b, ldg, M, Rp, want_colsum_b != 0, static_cast<hipStream_t>(stream));
}

extern "C" int cara_tskinny_partial2_r(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                                       const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b,
                                       int ldg, int M, int Rp, int rank, void* stream) {
  if (rank <= 0 || rank > Rp) return CARA_E_ARG;
  // (Xa == NULL: the second product alone)
  if ((Xa && !ts_args_ok(Xa, ldxa, Gta, ldg, slabs_a, M, K1a, Rp)) || !ts_args_ok(Xb, ldxb, Gtb, ldg, slabs_b, M, K1b, Rp)) return CARA_E_ARG;
  TsProblem a = ts_problem(Xa ? Xa : Xb, Xa ? ldxa : ldxb, Xa ? Gta : Gtb, Xa ? slabs_a : slabs_b, 0, M, Xa ? K1a : K1b, Rp);
  if (!Xa) a.nblk = 0;
  const TsProblem b = ts_problem(Xb, ldxb, Gtb, slabs_b, want_colsum_b, M, K1b, Rp);
  return ts_launch(a, b, ldg, M, Rp, want_colsum_b != 0, static_cast<hipStream_t>(stream), Rp == 32 && rank <= 16);
}

// library-internal (gemm8.h): cara_tskinny_partial2_r at rank <= 16 with the two-stage ring, for launches on a side stream UNDER a
// GEMM of the 160 x 256 x 64 tile (one workgroup per CU leaves 56 KiB of LDS and a third of the registers)
int cara_tskinny_partial2_small(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a, const void* Xb, int ldxb, const void* Gtb,
                                void* slabs_b, int K1b, int want_colsum_b, int ldg, int M, int Rp, int rank, void* stream) {
  if (rank <= 0 || rank > 16 || Rp != 32) return CARA_E_ARG;
  if ((Xa && !ts_args_ok(Xa, ldxa, Gta, ldg, slabs_a, M, K1a, Rp)) || !ts_args_ok(Xb, ldxb, Gtb, ldg, slabs_b, M, K1b, Rp)) return CARA_E_ARG;
  TsProblem a = ts_problem(Xa ? Xa : Xb, Xa ? ldxa : ldxb, Xa ? Gta : Gtb, Xa ? slabs_a : slabs_b, 0, M, Xa ? K1a : K1b, Rp);
  if (!Xa) a.nblk = 0;
  const TsProblem b = ts_problem(Xb, ldxb, Gtb, slabs_b, want_colsum_b, M, K1b, Rp);
  return ts_launch(a, b, ldg, M, Rp, want_colsum_b != 0, static_cast<hipStream_t>(stream), true, true);
}

extern "C" int cara_tskinny_reduce(const void* slabs, size_t slab_stride, float* D, float* colsum, int batch, int M,
                                   int K1, int Rp, void* stream) {
  if (!slabs || !D || batch <= 0 || M <= 0 || K1 <= 0 || (K1 % TS_COLS) || !(Rp == 32 || Rp == 64)) return CARA_E_ARG;
  const int total = K1 * Rp;
  hipLaunchKernelGGL(tskinny_reduce_kernel, dim3((total + 255) / 256, batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const char*>(slabs), slab_stride, D, colsum, K1, Rp, ts_chunks(M, K1));
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_tskinny_xtg(const void* X, int ldx, const void* Gt, int ldg, float* D, float* colsum,
                                void* slabs, int M, int K1, int Rp, void* stream) {
  if (!D) return CARA_E_ARG;
  const int st = cara_tskinny_partial(X, ldx, Gt, ldg, slabs, colsum != nullptr, M, K1, Rp, stream);
  if (st != CARA_OK) return st;
  return cara_tskinny_reduce(slabs, 0, D, colsum, 1, M, K1, Rp, stream);
}
