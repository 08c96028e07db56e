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
// Rounding W + dW to ONE bf16 costs what rounding W alone already costs (the error is set by W's ulp).
// HBM-bound: reads W, writes W_eff (+ its transpose for dX): 2 x 170 MB + 170 MB per step at ViT-B.
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
    Weff[e] = (bf16)((float)W[e] + keep_scale((unsigned)e, seed, lin, thresh, inv_keep) * d);
  }
}

// dVs[o,:] = sum_i m dW[o,i] U[i,:]: one wave per output row, lanes stride the columns
template <int RP>
__global__ __launch_bounds__(256) void contract_rows_kernel(const float* __restrict__ dW, const bf16* __restrict__ U, float* __restrict__ dVs,
                                                            int out, int in, unsigned seed, unsigned lin, unsigned thresh, float inv_keep) {
  const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= out) return;
  float acc[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) acc[r] = 0.f;
  for (int i = lane; i < in; i += 64) {
    const size_t e = (size_t)o * in + i;
    const float g = dW[e] * keep_scale((unsigned)e, seed, lin, thresh, inv_keep);
    if (g != 0.f) {
#pragma unroll
      for (int r = 0; r < RP; r += 8) {
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(U + (size_t)i * RP + r);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[r + k] += g * (float)u[k];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const float s = wave_sum(acc[r]);
    if (lane == 0) dVs[(size_t)o * RP + r] = s;
  }
}

// dU[i,:] = sum_o m dW[o,i] Vs[o,:]: a workgroup owns 64 columns; thread (column t%64, row group t/64) walks
// rows with stride 4 (coalesced 256-byte row segments), the 4 row groups are summed through LDS in fixed order
template <int RP>
__global__ __launch_bounds__(256) void contract_cols_kernel(const float* __restrict__ dW, const bf16* __restrict__ Vs, float* __restrict__ dU,
                                                            int out, int in, unsigned seed, unsigned lin, unsigned thresh, float inv_keep) {
  __shared__ float part[4][64][RP + 1];
  const int ic = threadIdx.x & 63, og = threadIdx.x >> 6, i = blockIdx.x * 64 + ic;
  float acc[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) acc[r] = 0.f;
  if (i < in) {
    for (int o = og; o < out; o += 4) {
      const size_t e = (size_t)o * in + i;
      const float g = dW[e] * keep_scale((unsigned)e, seed, lin, thresh, inv_keep);
      if (g != 0.f) {
#pragma unroll
        for (int r = 0; r < RP; r += 8) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(Vs + (size_t)o * RP + r);   // wave-uniform address: broadcast
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[r + k] += g * (float)v[k];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RP; ++r) part[og][ic][r] = acc[r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * RP; idx += 256) {
    const int r = idx % RP, c = idx / RP;
    if (blockIdx.x * 64 + c < in)
      dU[(size_t)(blockIdx.x * 64 + c) * RP + r] = ((part[0][c][r] + part[1][c][r]) + part[2][c][r]) + part[3][c][r];
  }
}

// column sums of a bf16 [M, ld] matrix (dc = sum_m dY): a workgroup owns 64 columns, 4 row groups, fixed-order sum
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int ld, int M, int N, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (c < N)
    for (int m = g; m < M; m += 4) s += (float)X[(size_t)m * ld + c];
  part[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < N) out[c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
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
  if (!W || !U || !Vs || !Weff || out <= 0 || in <= 0 || !(Rp == 32 || Rp == 64) || !mask_params(p, &thresh, &inv_keep)) return CARA_E_ARG;
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

extern "C" int cara_dropout_grad_contract(const float* dW, const void* U, const void* Vs, int Rp, int out, int in, float p, unsigned seed,
                                          unsigned linear_id, float* dU, float* dVs, void* stream) {
  unsigned thresh;
  float inv_keep;
  if (!dW || !U || !Vs || !dU || !dVs || out <= 0 || in <= 0 || !(Rp == 32 || Rp == 64) || !mask_params(p, &thresh, &inv_keep)) return CARA_E_ARG;
  if ((unsigned long long)out * in >= (1ull << 32)) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Rp == 32) {
    hipLaunchKernelGGL(contract_rows_kernel<32>, dim3((out + 3) / 4), dim3(256), 0, st, dW, (const bf16*)U, dVs, out, in, seed, linear_id, thresh, inv_keep);
    hipLaunchKernelGGL(contract_cols_kernel<32>, dim3((in + 63) / 64), dim3(256), 0, st, dW, (const bf16*)Vs, dU, out, in, seed, linear_id, thresh, inv_keep);
  } else {
    hipLaunchKernelGGL(contract_rows_kernel<64>, dim3((out + 3) / 4), dim3(256), 0, st, dW, (const bf16*)U, dVs, out, in, seed, linear_id, thresh, inv_keep);
    hipLaunchKernelGGL(contract_cols_kernel<64>, dim3((in + 63) / 64), dim3(256), 0, st, dW, (const bf16*)Vs, dU, out, in, seed, linear_id, thresh, inv_keep);
  }
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_colsum_bf16(const void* X, int ld, int M, int N, float* out, void* stream) {
  if (!X || !out || M <= 0 || N <= 0 || ld < N) return CARA_E_ARG;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16*)X, ld, M, N, out);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
