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

template <int NT>
__global__ __launch_bounds__(1024) void skinny_xu_sliced_kernel(const bf16* __restrict__ X, int ldx,
                                                                const bf16* __restrict__ Ut,
                                                                bf16* __restrict__ T, bf16* __restrict__ Tt,
                                                                int ldt, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);   // [nwaves][NT][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int Rp = NT * 16, KS = XU_KSLICE / 32;
  const int k0 = wave * XU_KSLICE + fq * 8;
  bf16x8 u[KS][NT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NT; ++j) u[ks][j] = *reinterpret_cast<const bf16x8*>(Ut + (size_t)(j * 16 + fr) * K + k0 + ks * 32);
  const int mblk = blockIdx.x * (16 * XU_GROUPS);
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
    const int m0 = mblk + g * 16;
    if (g + 1 < XU_GROUPS) {
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
constexpr int TS_COLS = 64;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__host__ __device__ inline int ts_chunks(int M, int K1) {
  const int colblocks = K1 / TS_COLS;
  int c = (256 + colblocks - 1) / colblocks;  // ~1 block per CU
  const int steps = (M + 31) / 32;
  const int maxc = (steps + 7) / 8;  // at least 8 steps (2 per wave) per block
  if (c > maxc) c = maxc;
  if (c < 1) c = 1;
  return c;
}

template <int NT>
struct TsRing {
  static constexpr int X_BYTES = 32 * TS_COLS * 2;   // 4 KiB : 4 pieces of 8 rows x 128 B
  static constexpr int G_BYTES = NT * 16 * 32 * 2;   // Rp rows x 64 B : NT pieces of 16 rows
  static constexpr int STAGE = X_BYTES + G_BYTES;
  static constexpr int PIECES = 4 + NT;
  static constexpr int WAVE_BYTES = 3 * STAGE;
};

template <int NT>
__device__ __forceinline__ void ts_issue(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Gt,
                                         int ldg, int i0, int m0, int M, char* stage, int lane) {
  // X tile: piece q = rows 8q..8q+7; lane -> row 8q + lane/8, 16-B slot lane%8 holding global
  // chunk (lane%8) ^ q  (reader of row group fq = q reads chunk c ^ fq: bank-conflict free)
  // ldx < 0: X is K-panel-major, [K1/32][-ldx rows][32]: chunk cg of the 64 columns = panel i0/32 + cg/4, 16-B piece cg%4
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int m = m0 + q * 8 + (lane >> 3);
    m = m < M ? m : M - 1;
    const int cg = (lane & 7) ^ q;
    const bf16* src = ldx < 0 ? X + ((size_t)((i0 >> 5) + (cg >> 2)) * (size_t)(-ldx) + m) * 32 + (cg & 3) * 8
                              : X + (size_t)m * ldx + i0 + cg * 8;
    glds16(src, stage + q * 1024);
  }
  // Gt tile: [Rp][32 m] bf16, 64-B rows; piece p = rows 16p..16p+15, lane -> row lane/4, chunk lane%4
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    const int r = p * 16 + (lane >> 2);
    glds16(Gt + (size_t)r * ldg + m0 + (lane & 3) * 8, stage + TsRing<NT>::X_BYTES + p * 1024);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt");
}

// one product D = X^T G; two of them (dU and dVs of one linear) share a launch
struct TsProblem {
  const bf16* X; const bf16* Gt;
  float* slabs; float* cs_slabs;     // cs_slabs == nullptr: no column sums wanted for this problem
  int ldx, K1, nchunks, nblk;
};

template <int NT, bool COLSUM>
__global__ __launch_bounds__(256) void tskinny_kernel(const TsProblem p0, const TsProblem p1, int ldg, int M) {
  using R = TsRing<NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const bool second = (int)blockIdx.x >= p0.nblk;
  const TsProblem& P = second ? p1 : p0;
  const int bid = second ? blockIdx.x - p0.nblk : blockIdx.x;
  const bf16* __restrict__ X = P.X;
  const bf16* __restrict__ Gt = P.Gt;
  float* __restrict__ slabs = P.slabs;
  float* __restrict__ cs_slabs = P.cs_slabs;
  const int ldx = P.ldx, K1 = P.K1, nchunks = P.nchunks;
  const bool want_cs = COLSUM && cs_slabs != nullptr;
  const int colblocks = K1 / TS_COLS;
  const int cb = bid % colblocks, chunk = bid / colblocks;
  const int i0 = cb * TS_COLS;
  const int steps = (M + 31) / 32;
  const int s_begin = (int)((long)steps * chunk / nchunks), s_end = (int)((long)steps * (chunk + 1) / nchunks);
  char* ring = smem + wave * R::WAVE_BYTES;

  f32x4 acc[NT][4];  // [r-tile][i-tile]
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  // this wave's steps: s_begin + wave, +4, ...
  const int first = s_begin + wave;
  const int nmine = first < s_end ? (s_end - first + 3) / 4 : 0;
  if (nmine > 0) ts_issue<NT>(X, ldx, Gt, ldg, i0, first * 32, M, ring, lane);
  if (nmine > 1) ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4) * 32, M, ring + R::STAGE, lane);
  for (int t = 0; t < nmine; ++t) {
    const int slot = t % 3;
    if (t + 2 < nmine) {
      ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4 * (t + 2)) * 32, M, ring + ((t + 2) % 3) * R::STAGE, lane);
      wait_vmcnt<2 * R::PIECES>();
    } else if (t + 1 < nmine) {
      wait_vmcnt<R::PIECES>();
    } else {
      wait_vmcnt<0>();
    }
    const char* sx = ring + slot * R::STAGE;
    const char* sg = sx + R::X_BYTES;
    bf16x8 a[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
      a[rt] = *reinterpret_cast<const bf16x8*>(sg + (rt * 16 + fr) * 64 + fq * 16);
    const int mrow = (first + 4 * t) * 32 + fq * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // B fragment: X[m = 8 fq + j][i = it*16 + fr], j = 0..7 -- k runs over the image ROWS, so it is
      // taken with the transposing LDS read: the 16-lane group fq reads the 4x16 blocks of rows
      // 8fq..8fq+3 and 8fq+4..8fq+7, columns it*16..it*16+15; lane 4q+p supplies (row q, cols 4p..4p+3),
      // lane fr receives column fr (pinned by tests/test_kernels_gpu.py::test_transposing_lds_read_semantics)
      const int trow = fq * 8 + (fr >> 2);
      const int toff = ((((it * 2 + ((fr & 3) >> 1)) ^ fq) << 4) | ((fr & 1) << 3));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sx + trow * 128 + toff));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sx + (trow + 4) * 128 + toff));
      const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
      const bf16x8 b = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      if constexpr (COLSUM) {
        if (want_cs) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[it] += (mrow + j < M) ? (float)b[j] : 0.f;
        }
      }
#pragma unroll
      for (int rt = 0; rt < NT; ++rt)
        acc[rt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], b, acc[rt][it], 0, 0, 0);
    }
  }
  // ---- combine the 4 waves through LDS (ring memory is dead now) ----
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [wave][NT*4][64]
#pragma unroll
  for (int rt = 0; rt < NT; ++rt)
#pragma unroll
    for (int it = 0; it < 4; ++it) red[(wave * NT * 4 + rt * 4 + it) * 64 + lane] = acc[rt][it];
  float* cred = reinterpret_cast<float*>(smem + 4 * NT * 4 * 64 * 16);  // [wave][4 it][64 lanes]
  if constexpr (COLSUM) {
#pragma unroll
    for (int it = 0; it < 4; ++it) cred[(wave * 4 + it) * 64 + lane] = csum[it];
  }
  __syncthreads();
  float* slab = slabs + (size_t)bid * TS_COLS * (NT * 16);
  for (int t = wave; t < NT * 4; t += 4) {
    f32x4 s = red[t * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = red[(w * NT * 4 + t) * 64 + lane];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int rt = t >> 2, it = t & 3;
    // C layout: row (= r) = rt*16 + fq*4 + reg, col (= i) = it*16 + fr  ->  slab[i][r..r+3]
    *reinterpret_cast<f32x4*>(slab + (size_t)(it * 16 + fr) * (NT * 16) + rt * 16 + fq * 4) = s;
  }
  if constexpr (COLSUM) {
    if (want_cs && tid < TS_COLS) {
      const int it = tid >> 4, f = tid & 15;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += cred[(w * 4 + it) * 64 + q * 16 + f];
      cs_slabs[(size_t)bid * TS_COLS + tid] = s;
    }
  }
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
__global__ void tskinny_reduce_many_kernel(const TsReduceTable t) {
  const cara_ts_reduce& q = t.p[blockIdx.z];
  if ((int)blockIdx.y >= q.batch) return;
  const int K1 = q.K1, Rp = q.Rp, nchunks = t.nchunks[blockIdx.z];
  const int colblocks = K1 / TS_COLS;
  const int nblk = colblocks * nchunks;
  const float* slabs = reinterpret_cast<const float*>(static_cast<const char*>(q.slabs) + (size_t)blockIdx.y * q.slab_stride);
  const float* cs_slabs = slabs + (size_t)nblk * TS_COLS * Rp;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = K1 * Rp;
  if (idx < total) {
    const int i = idx / Rp, r = idx - i * Rp;
    const int cb = i / TS_COLS, il = i - cb * TS_COLS;
    const float s = strided_sum4(slabs + ((size_t)cb * TS_COLS + il) * Rp + r, (size_t)colblocks * TS_COLS * Rp, nchunks);
    q.D[(size_t)blockIdx.y * total + idx] = s;
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
    t.p[i] = q;
    t.nchunks[i] = ts_chunks(q.M, q.K1);
    const int blocks = (q.K1 * q.Rp + 255) / 256;
    maxblocks = blocks > maxblocks ? blocks : maxblocks;
    maxbatch = q.batch > maxbatch ? q.batch : maxbatch;
  }
  hipLaunchKernelGGL(tskinny_reduce_many_kernel, dim3(maxblocks, maxbatch, n), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_skinny_xu(const void* X, int ldx, const void* Ut, void* T, void* Tt, int ldt,
                              int M, int K, int Rp, void* stream) {
  if (!X || !Ut || !T || M <= 0 || K <= 0 || (K & 31)) return CARA_E_ARG;
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
  // (Rp = 64 would need 96 VGPRs of Ut fragments per wave: over the 128-VGPR budget of a
  // 1024-thread workgroup, so rank > 32 stays on v1)
  if (Rp == 32 && K % XU_KSLICE == 0 && K / XU_KSLICE <= 16 && M >= 64) {
    const int nw = K / XU_KSLICE;
    const dim3 g2((M + 16 * XU_GROUPS - 1) / (16 * XU_GROUPS)), b2(nw * 64);
    const size_t lds = (size_t)nw * 2 * 64 * sizeof(f32x4);
    hipLaunchKernelGGL(skinny_xu_sliced_kernel<2>, g2, b2, lds, st, x, ldx, u, (bf16*)T, (bf16*)Tt, ldt, M, K);
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
  const size_t nblk = (size_t)(K1 / TS_COLS) * ts_chunks(M, K1);
  return nblk * TS_COLS * Rp * sizeof(float) + nblk * TS_COLS * sizeof(float);
}

namespace {
bool ts_args_ok(const void* X, int ldx, const void* Gt, int ldg, void* slabs, int M, int K1, int Rp) {
  if (!X || !Gt || !slabs || M <= 0 || K1 <= 0 || (K1 % TS_COLS)) return false;
  if (ldx < 0 ? -ldx < M : ((ldx & 7) || ldx < K1)) return false;   // ldx < 0: K-panel-major X, -ldx rows per panel
  // Gt rows must be readable (and zero) up to the next multiple of 32 rows of M
  if ((ldg & 7) || ldg < ((M + 31) / 32) * 32) return false;
  return Rp == 32 || Rp == 64;
}
TsProblem ts_problem(const void* X, int ldx, const void* Gt, void* slabs, int want_colsum, int M, int K1, int Rp) {
  TsProblem p;
  p.X = (const bf16*)X; p.Gt = (const bf16*)Gt; p.ldx = ldx; p.K1 = K1;
  p.nchunks = ts_chunks(M, K1);
  p.nblk = (K1 / TS_COLS) * p.nchunks;
  p.slabs = static_cast<float*>(slabs);
  p.cs_slabs = want_colsum ? p.slabs + (size_t)p.nblk * TS_COLS * Rp : nullptr;
  return p;
}
int ts_launch(const TsProblem& a, const TsProblem& b, int ldg, int M, int Rp, bool any_cs, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<4>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<4>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<2>::WAVE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tskinny_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TsRing<2>::WAVE_BYTES);
    attr_set = true;
  }
  const dim3 grid(a.nblk + b.nblk), block(256);
  if (Rp == 32) {
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
