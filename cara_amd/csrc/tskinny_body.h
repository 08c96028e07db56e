// The transposed skinny product D[K1, Rp] = X[M, K1]^T . G[M, Rp] (dU = X^T G', dVs = dY^T T) as DEVICE code, shared
// by its own kernel (skinny.hip: a private 3-deep LDS ring per wave) and by the dX GEMM that carries the products
// of its linear in the same launch (gemm.hip, gemm32_ts_kernel: one stage per wave, the GEMM's LDS budget).
// See skinny.hip for the algorithm.
#pragma once
#include "common.h"

namespace {

constexpr int TS_COLS = 64;
#ifndef TS_TARGET_BLOCKS
#define TS_TARGET_BLOCKS 256
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;

__host__ __device__ inline int ts_chunks(int M, int K1) {
  const int colblocks = K1 / TS_COLS;
  int c = (TS_TARGET_BLOCKS + colblocks - 1) / colblocks;  // ~TS_TARGET_BLOCKS / 256 blocks per CU
  const int steps = (M + 31) / 32;
  const int maxc = (steps + 7) / 8;  // at least 8 steps (2 per wave) per block
  if (c > maxc) c = maxc;
  if (c < 1) c = 1;
  return c;
}

template <int NT, int NSTAGE = 3>
struct TsRing {
  static constexpr int X_BYTES = 32 * TS_COLS * 2;   // 4 KiB : 4 pieces of 8 rows x 128 B
  static constexpr int G_BYTES = NT * 16 * 32 * 2;   // Rp rows x 64 B : NT pieces of 16 rows
  static constexpr int STAGE = X_BYTES + G_BYTES;
  static constexpr int PIECES = 4 + NT;
  static constexpr int WAVE_BYTES = NSTAGE * STAGE;
  // the end-of-block combine runs in passes of 8 accumulator tiles (two r-tiles x four i-tiles): [4 waves][8][64] f32x4
  // + [4 waves][4][64] floats, whatever NT (Rp = 64 as one pass would be 68 KiB: two workgroups per CU for the GEMM
  // the products ride in; NT = 1 -- rank <= 16, one r-tile -- uses half of a pass)
  static constexpr int COMBINE_BYTES = 4 * 8 * 64 * 16 + 4 * 4 * 64 * 4;
  static constexpr int BLOCK_BYTES = 4 * WAVE_BYTES > COMBINE_BYTES ? 4 * WAVE_BYTES : COMBINE_BYTES;
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
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
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

// one block's share; `block` = its index among the p0.nblk + p1.nblk blocks of the pair.  NSTAGE = 3: two K steps
// of prefetch per wave; NSTAGE = 1: load, wait, multiply (enough when the CU holds many other waves)
// tid_in / active: a workgroup of more than 256 threads runs one block per 256 threads (the 512-thread GEMM tile of gemm8.hip
// that carries the products); a surplus group passes active = false: it keeps the barrier count and touches no memory
template <int NT, bool COLSUM, int NSTAGE>
__device__ __forceinline__ void tskinny_body(const TsProblem& p0, const TsProblem& p1, const int ldg, const int M, const int block,
                                             char* smem, const int tid_in = -1, const bool active = true) {
  using R = TsRing<NT, NSTAGE>;
  const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const bool second = block >= p0.nblk;
  const TsProblem& P = second ? p1 : p0;
  const int bid = second ? block - p0.nblk : block;
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
  const int nmine = (active && first < s_end) ? (s_end - first + 3) / 4 : 0;
  if constexpr (NSTAGE == 3) {
    if (nmine > 0) ts_issue<NT>(X, ldx, Gt, ldg, i0, first * 32, M, ring, lane);
    if (nmine > 1) ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4) * 32, M, ring + R::STAGE, lane);
  }
  if constexpr (NSTAGE == 2) {
    if (nmine > 0) ts_issue<NT>(X, ldx, Gt, ldg, i0, first * 32, M, ring, lane);
  }
  for (int t = 0; t < nmine; ++t) {
    int slot = 0;
    if constexpr (NSTAGE == 2) {   // one K step of prefetch (40 KiB per block: fits beside a 160 x 256 x 64 GEMM tile on its CU)
      slot = t & 1;
      if (t + 1 < nmine) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the other stage's LDS reads (step t - 1) are done
        ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4 * (t + 1)) * 32, M, ring + ((t + 1) & 1) * R::STAGE, lane);
        wait_vmcnt<R::PIECES>();
      } else {
        wait_vmcnt<0>();
      }
    } else if constexpr (NSTAGE == 3) {
      slot = t % 3;
      if (t + 2 < nmine) {
        ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4 * (t + 2)) * 32, M, ring + ((t + 2) % 3) * R::STAGE, lane);
        wait_vmcnt<2 * R::PIECES>();
      } else if (t + 1 < nmine) {
        wait_vmcnt<R::PIECES>();
      } else {
        wait_vmcnt<0>();
      }
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous step's LDS reads are done: the stage is free
      ts_issue<NT>(X, ldx, Gt, ldg, i0, (first + 4 * t) * 32, M, ring, lane);
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
  // ---- combine the 4 waves through LDS (ring memory is dead now), two r-tiles (8 accumulator tiles) per pass ----
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [wave][8][64]
  float* cred = reinterpret_cast<float*>(smem + 4 * 8 * 64 * 16);  // [wave][4 it][64 lanes]
  float* slab = slabs + (size_t)bid * TS_COLS * (NT * 16);
  constexpr int RPP = NT >= 2 ? 2 : 1;   // r-tiles per pass
#pragma unroll
  for (int h = 0; h < NT / RPP; ++h) {
    if (h) __syncthreads();   // the previous pass's sums have been read
#pragma unroll
    for (int r2 = 0; r2 < RPP; ++r2)
#pragma unroll
      for (int it = 0; it < 4; ++it) red[(wave * 8 + r2 * 4 + it) * 64 + lane] = acc[h * RPP + r2][it];
    if constexpr (COLSUM) {
      if (h == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) cred[(wave * 4 + it) * 64 + lane] = csum[it];
      }
    }
    __syncthreads();
    for (int t = wave; active && t < 4 * RPP; t += 4) {
      f32x4 s = red[t * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 v = red[(w * 8 + t) * 64 + lane];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      const int rt = h * RPP + (t >> 2), it = t & 3;
      // C layout: row (= r) = rt*16 + fq*4 + reg, col (= i) = it*16 + fr  ->  slab[i][r..r+3]
      *reinterpret_cast<f32x4*>(slab + (size_t)(it * 16 + fr) * (NT * 16) + rt * 16 + fq * 4) = s;
    }
  }
  if constexpr (COLSUM) {
    if (active && want_cs && tid < TS_COLS) {
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

// host side: argument checks and the per-product descriptor
inline bool ts_args_ok(const void* X, int ldx, const void* Gt, int ldg, void* slabs, int M, int K1, int Rp) {
  if (!X || !Gt || !slabs || M <= 0 || K1 <= 0 || (K1 % TS_COLS)) return false;
  if (ldx < 0 ? -ldx < M : ((ldx & 7) || ldx < K1)) return false;   // ldx < 0: K-panel-major X, -ldx rows per panel
  // Gt rows must be readable (and zero) up to the next multiple of 32 rows of M
  if ((ldg & 7) || ldg < ((M + 31) / 32) * 32) return false;
  return Rp == 32 || Rp == 64;
}
// (the column sums sit behind nblk slabs of the FULL width Rp whatever the rank: the place cara_tskinny_reduce* looks for them)
inline TsProblem ts_problem(const void* X, int ldx, const void* Gt, void* slabs, int want_colsum, int M, int K1, int Rp) {
  TsProblem p;
  p.X = (const bf16*)X; p.Gt = (const bf16*)Gt; p.ldx = ldx; p.K1 = K1;
  p.nchunks = ts_chunks(M, K1);
  p.nblk = (K1 / TS_COLS) * p.nchunks;
  p.slabs = static_cast<float*>(slabs);
  p.cs_slabs = want_colsum ? p.slabs + (size_t)p.nblk * TS_COLS * Rp : nullptr;
  return p;
}
}  // namespace
