// Exact weight-space dropout mode (gfx950): the reference's TRAIN-mode arithmetic.
//
// In train mode the reference applies nn.Dropout(0.1) to every materialised adapter tensor before using it
// (/root/reference/src/cara/cara.py:35 qkv, :57 proj, :81 fc1, :92 fc2):  y = x W^T + b + s (x dp(dW)^T + c).
// A mask on the ELEMENTS of dW does not factor through U g V^T, so this mode does what the reference does --
// it materialises the masked adapter -- but once per step and merged into the frozen weight:
//     W_eff[o,i] = bf16( W[o,i] + keep(o,i)/(1-p) * sum_r Vs[o,r] U[i,r] )          (cara_materialize_merge)
// after which forward and dX are PLAIN GEMMs on W_eff (no skinny products, no K extension), and the adapter
// gradients come from the dense weight gradient dW = dY^T X the reference also forms:
//     dVs[o,r] = sum_i keep(o,i)/(1-p) dW[o,i] U[i,r],   dU[i,r] = sum_o keep(o,i)/(1-p) dW[o,i] Vs[o,r]
// (cara_dropout_grad_contract), which are exactly the per-layer quantities the factored path hands to
// cara_factor_grad_reduce.  keep(o,i) is a counter-based hash of (seed, linear id, o*in + i): nothing is stored,
// the backward regenerates the mask, and tests regenerate it on the CPU (tests/test_exact_dropout.py).
// Rounding W + dW to ONE bf16 loses every adapter element below half an ulp of W (|W| ~ 0.02: 6e-5) -- with the
// reference's zero-initialised A2 / P2 that is the whole adapter for a long stretch of training.  The whole-model
// path therefore keeps the two apart: with W == NULL this kernel writes only the masked delta
//     Dm[o,i] = bf16( keep(o,i)/(1-p) * sum_r Vs[o,r] U[i,r] )                         (full relative precision)
// and the GEMMs run y = x W^T + x Dm^T as two products accumulated in fp32 (cara_gemm_args::B3).  The merged form
// stays for the eval-time merge (p = 0) of a trained adapter into the backbone.
// HBM-bound: writes Dm (+ its transpose for dX): 2 x 170 MB per step at ViT-B.
#include "common.h"

namespace {

// lowbias32 finaliser over (element index, seed, linear id); bit-exactly mirrored in tests (numpy uint32)
__host__ __device__ __forceinline__ unsigned keep_hash(unsigned idx, unsigned seed, unsigned lin) {
  unsigned h = idx * 0x9E3779B1u ^ (seed + lin * 0x85EBCA77u);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
// keep with probability 1 - p: compare the top 24 bits with p * 2^24
__device__ __forceinline__ float keep_scale(unsigned idx, unsigned seed, unsigned lin, unsigned thresh, float inv_keep) {
  return (keep_hash(idx, seed, lin) >> 8) >= thresh ? inv_keep : 0.f;
}

// one workgroup: 64 output rows x 64 input columns; thread t: row t/4, 16 consecutive columns
template <int RP>
__global__ __launch_bounds__(256) void merge_kernel(const bf16* __restrict__ W, const bf16* __restrict__ U, const bf16* __restrict__ Vs,
                                                    bf16* __restrict__ Weff, int out, int in, unsigned seed, unsigned lin,
                                                    unsigned thresh, float inv_keep) {
  __shared__ float Us[64][RP + 1];
  const int o0 = blockIdx.y * 64, i0 = blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * RP; idx += 256) {
    const int r = idx % RP, i = idx / RP;
    Us[i][r] = (i0 + i < in) ? (float)U[(size_t)(i0 + i) * RP + r] : 0.f;
  }
  __syncthreads();
  const int o = o0 + (threadIdx.x >> 2), ic = (threadIdx.x & 3) * 16;
  if (o >= out) return;
  float v[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) v[r] = (float)Vs[(size_t)o * RP + r];
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int i = i0 + ic + j;
    if (i >= in) break;
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < RP; ++r) d += v[r] * Us[ic + j][r];
    const size_t e = (size_t)o * in + i;
    Weff[e] = (bf16)((W ? (float)W[e] : 0.f) + keep_scale((unsigned)e, seed, lin, thresh, inv_keep) * d);
  }
}

// dVs[o,:] = sum_i m dW[o,i] U[i,:]: one wave per output row, lanes stride the columns
// dW arrives as `nslab` split-K partial slabs (fp32 [out,in] each, slab_stride floats apart) that are summed here
__device__ __forceinline__ float slab_sum(const float* __restrict__ dW, size_t e, int nslab, size_t slab_stride) {
  float g = dW[e];
  for (int sidx = 1; sidx < nslab; ++sidx) g += dW[e + sidx * slab_stride];
  return g;
}

// Both contractions in ONE pass over dW.  A workgroup owns a 64 x 64 tile of dW: it sums the split-K slabs, applies the
// regenerated mask and leaves the masked fp32 tile in LDS next to the tile's 64 rows of U and of Vs; every thread then
// forms 8 outputs of each contraction (a row of the tile against 8 columns of U, a column against 8 columns of Vs) and the
// tile's partial sums go to scratch -- dVs partials [in / 64][out][RP], dU partials [out / 64][in][RP] -- which
// reduce_chunks_kernel adds in fixed order.  (The two separate passes read every slab twice, hashed every element twice and
// ran at 1 TB/s: 72 us per linear at the headline shape, a seventh of the exact-mode step.)
template <int RP>
__global__ __launch_bounds__(256) void contract_tile_kernel(const float* __restrict__ dW, int nslab, size_t slab_stride,
                                                            const bf16* __restrict__ U, const bf16* __restrict__ Vs,
                                                            float* __restrict__ pV, float* __restrict__ pU, int out, int in,
                                                            unsigned seed, unsigned lin, unsigned thresh, float inv_keep) {
  __shared__ float g[64][65];
  __shared__ __attribute__((aligned(16))) float us[64][RP + 4];
  __shared__ __attribute__((aligned(16))) float vs[64][RP + 4];
  const int t = threadIdx.x;
  const int i0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
  // the masked tile: thread -> row idx / 16, four consecutive columns
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = t + k * 256, r = idx >> 4, c = (idx & 15) * 4;
    const int o = o0 + r, i = i0 + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < out) {
      const size_t e = (size_t)o * in + i;
      if (i + 4 <= in && (in & 3) == 0) {
        f32x4 a = *reinterpret_cast<const f32x4*>(dW + e);
        for (int sidx = 1; sidx < nslab; ++sidx) a += *reinterpret_cast<const f32x4*>(dW + e + sidx * slab_stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a[j] * keep_scale((unsigned)(e + j), seed, lin, thresh, inv_keep);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i + j < in) v[j] = slab_sum(dW, e + j, nslab, slab_stride) * keep_scale((unsigned)(e + j), seed, lin, thresh, inv_keep);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) g[r][c + j] = v[j];
  }
  for (int idx = t; idx < 64 * (RP / 8); idx += 256) {
    const int r = idx / (RP / 8), c = (idx % (RP / 8)) * 8;
    bf16x8 a = {}, b = {};
    if (i0 + r < in) a = *reinterpret_cast<const bf16x8*>(U + (size_t)(i0 + r) * RP + c);
    if (o0 + r < out) b = *reinterpret_cast<const bf16x8*>(Vs + (size_t)(o0 + r) * RP + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      us[r][c + j] = (float)a[j];
      vs[r][c + j] = (float)b[j];
    }
  }
  __syncthreads();
  // outputs: thread -> (row / column q, 8 of the RP columns starting at r0), RP / 8 threads per q, 256 / (RP / 8) q per pass
  constexpr int TPQ = RP / 8, QPP = 256 / TPQ;
  const int r0 = (t % TPQ) * 8;
#pragma unroll
  for (int pass = 0; pass < 64 / QPP; ++pass) {
    const int q = pass * QPP + t / TPQ;
    float av[8] = {}, au[8] = {};
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      const float gv = g[q][k], gu = g[k][q];      // dVs: row q against U rows k; dU: column q against Vs rows k
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(&us[k][r0]), u1 = *reinterpret_cast<const f32x4*>(&us[k][r0 + 4]);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&vs[k][r0]), v1 = *reinterpret_cast<const f32x4*>(&vs[k][r0 + 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[j] += gv * u0[j]; av[4 + j] += gv * u1[j];
        au[j] += gu * v0[j]; au[4 + j] += gu * v1[j];
      }
    }
    if (o0 + q < out) {
      float* d = pV + ((size_t)blockIdx.x * out + o0 + q) * RP + r0;
      *reinterpret_cast<f32x4*>(d) = f32x4{av[0], av[1], av[2], av[3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{av[4], av[5], av[6], av[7]};
    }
    if (i0 + q < in) {
      float* d = pU + ((size_t)blockIdx.y * in + i0 + q) * RP + r0;
      *reinterpret_cast<f32x4*>(d) = f32x4{au[0], au[1], au[2], au[3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{au[4], au[5], au[6], au[7]};
    }
  }
}

// out[j] = sum over `chunks` partial arrays of n floats each (fixed order)
// (four interleaved partial sums keep four loads in flight per thread; the order is fixed: (s0 + s1) + (s2 + s3))
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ scratch, int chunks, size_t n, float* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float* p = scratch + j;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    s0 += p[(size_t)c * n];
    s1 += p[(size_t)(c + 1) * n];
    s2 += p[(size_t)(c + 2) * n];
    s3 += p[(size_t)(c + 3) * n];
  }
  if (c < chunks) s0 += p[(size_t)c * n];
  if (c + 1 < chunks) s1 += p[(size_t)(c + 1) * n];
  if (c + 2 < chunks) s2 += p[(size_t)(c + 2) * n];
  out[j] = (s0 + s1) + (s2 + s3);
}

// column sums of a bf16 [M, ld] matrix (dc = sum_m dY): workgroup (x, y) owns 256 columns and the y-th row chunk; a wave
// reads 256 columns of two rows per instruction (16 bytes per lane: lanes 0..31 one row, 32..63 the next); the two row
// halves and the workgroup's four waves are summed in fixed order into scratch [chunk][N]; reduce_chunks_kernel
// finishes.  (The first form read two bytes per lane: 32 us for 19 - 77 MB.)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int ld, int M, int N, float* __restrict__ scratch) {
  __shared__ float part[4][2][256];   // [wave][row of the instruction's pair][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = (lane & 31) * 8, rp = lane >> 5;           // 32 lanes x 8 columns = 256 columns; two rows per instruction
  const int c0 = blockIdx.x * 256 + cl;
  const int per = (M + gridDim.y - 1) / gridDim.y, m_begin = blockIdx.y * per, m_end = min(M, m_begin + per);
  float s[8] = {};
  const bool vec = c0 + 8 <= N && (ld & 7) == 0;
  for (int m = m_begin + wave * 2 + rp; m < m_end; m += 8) {
    if (vec) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)m * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c0 + j < N) s[j] += (float)X[(size_t)m * ld + c0 + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[wave][rp][cl + j] = s[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float tsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) tsum += part[w][0][threadIdx.x] + part[w][1][threadIdx.x];
    scratch[(size_t)blockIdx.y * N + c] = tsum;
  }
}

bool mask_params(float p, unsigned* thresh, float* inv_keep) {
  if (!(p >= 0.f && p < 1.f)) return false;
  *thresh = (unsigned)((double)p * 16777216.0);
  *inv_keep = 1.0f / (1.0f - p);
  return true;
}

}  // namespace

extern "C" unsigned cara_weight_dropout_hash(unsigned idx, unsigned seed, unsigned linear_id) { return keep_hash(idx, seed, linear_id); }

extern "C" int cara_materialize_merge(const void* W, const void* U, const void* Vs, int Rp, int out, int in, float p, unsigned seed,
                                      unsigned linear_id, void* Weff, void* stream) {
  unsigned thresh;
  float inv_keep;
  if (!U || !Vs || !Weff || out <= 0 || in <= 0 || !(Rp == 32 || Rp == 64) || !mask_params(p, &thresh, &inv_keep)) return CARA_E_ARG;
  if ((unsigned long long)out * in >= (1ull << 32)) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((in + 63) / 64, (out + 63) / 64);
  if (Rp == 32)
    hipLaunchKernelGGL(merge_kernel<32>, grid, dim3(256), 0, st, (const bf16*)W, (const bf16*)U, (const bf16*)Vs, (bf16*)Weff, out, in, seed,
                       linear_id, thresh, inv_keep);
  else
    hipLaunchKernelGGL(merge_kernel<64>, grid, dim3(256), 0, st, (const bf16*)W, (const bf16*)U, (const bf16*)Vs, (bf16*)Weff, out, in, seed,
                       linear_id, thresh, inv_keep);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// dVs partials [ceil(in / 64)][out][Rp] + dU partials [ceil(out / 64)][in][Rp], fp32
extern "C" size_t cara_dropout_grad_scratch_bytes(int out, int in, int Rp) {
  if (out <= 0 || in <= 0 || Rp <= 0) return 0;
  return ((size_t)((in + 63) / 64) * out + (size_t)((out + 63) / 64) * in) * Rp * sizeof(float);
}

extern "C" int cara_dropout_grad_contract(const float* dW, int nslab, size_t slab_stride, const void* U, const void* Vs, int Rp, int out,
                                          int in, float p, unsigned seed, unsigned linear_id, float* dU, float* dVs, void* scratch,
                                          void* stream) {
  unsigned thresh;
  float inv_keep;
  if (!dW || !U || !Vs || !dU || !dVs || !scratch || nslab <= 0 || out <= 0 || in <= 0 || !(Rp == 32 || Rp == 64) ||
      !mask_params(p, &thresh, &inv_keep))
    return CARA_E_ARG;
  if ((unsigned long long)out * in >= (1ull << 32)) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ti = (in + 63) / 64, to = (out + 63) / 64;
  float* pV = static_cast<float*>(scratch);               // [ti][out][Rp]
  float* pU = pV + (size_t)ti * out * Rp;                  // [to][in][Rp]
  const dim3 grid(ti, to);
  if (Rp == 32)
    hipLaunchKernelGGL(contract_tile_kernel<32>, grid, dim3(256), 0, st, dW, nslab, slab_stride, (const bf16*)U, (const bf16*)Vs, pV, pU, out, in, seed, linear_id, thresh, inv_keep);
  else
    hipLaunchKernelGGL(contract_tile_kernel<64>, grid, dim3(256), 0, st, dW, nslab, slab_stride, (const bf16*)U, (const bf16*)Vs, pV, pU, out, in, seed, linear_id, thresh, inv_keep);
  const size_t nv = (size_t)out * Rp, nu = (size_t)in * Rp;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, pV, ti, nv, dVs);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, pU, to, nu, dU);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" size_t cara_colsum_scratch_bytes(int N) { return (size_t)256 * N * sizeof(float); }

extern "C" int cara_colsum_bf16(const void* X, int ld, int M, int N, float* out, void* scratch, void* stream) {
  if (!X || !out || !scratch || M <= 0 || N <= 0 || ld < N) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // enough blocks to fill the chip: (N / 256 column groups) x chunks >= ~512; scratch holds up to 256 chunk rows
  int chunks = M >= 256 ? (512 * 256 + N - 1) / N : 1;
  chunks = chunks > 256 ? 256 : chunks;
  while (chunks > 1 && M / chunks < 16) chunks >>= 1;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, st, (const bf16*)X, ld, M, N, static_cast<float*>(scratch));
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((N + 255) / 256), dim3(256), 0, st, static_cast<const float*>(scratch), chunks, (size_t)N, out);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
