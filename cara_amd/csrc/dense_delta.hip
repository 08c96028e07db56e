// The order-2 QKV tensorisation of image_classification/dim_experiment.py (cp_length 2, :203-207, :293-297) on gfx950.
//
//   tensor_attn[k, e * dim + o] = sum_r R1[r] A1[3 l + k, r] A2[e * dim + o, r]        CP_A2 is [dim * dim, rank]
//   qkv_delta[k, b, n, o]       = sum_e x[b, n, e] tensor_attn[k, e, o]
//
// i.e. every projection gets a sum of `rank` DENSE dim x dim matrices: not low-rank in (in, out), so none of the factored
// kernels applies to the QKV linear of this order (proj / fc1 / fc2 keep them).  It runs instead in the dense-delta form the
// exact weight-dropout mode already uses: the scaled delta is materialised once per step next to the frozen weight,
//     Dm[l][k dim + o][e] = bf16( s * tensor_attn[k, e, o] )        (and its transpose, for dX)
// forward and dX are two products on the same operand accumulated in fp32 (cara_gemm_args::B3), and the gradients of
// CP_A1 / CP_A2 / CP_R1 come from the dense dD[l][k][e, o] = sum_m x[m, e] dY_k[m, o] (cara_gemm_tn_f32 on the row-major
// activations), contracted here:
//     dA2[p, r]      = s sum_{l,k} R1[r] A1[3l+k, r] dD[l][k][p]
//     dA1[3l+k, r]   = s R1[r] S[l,k][r],   dR1[r] = s sum_{l,k} A1[3l+k, r] S[l,k][r],   S[l,k][r] = sum_p dD[l][k][p] A2[p, r]
// Streaming kernels over the dim * dim rows of CP_A2 (37.7 MB at rank 16): fixed summation orders, no atomics.
#include "common.h"

namespace {

constexpr int DD_MAXLK = 192;   // 3 * depth <= 192 (depth <= 64)

// Dm / Dmt of all layers: grid (dim / 32 tiles of o, dim / 32 tiles of e, depth); thread -> four (e, o) pairs of the tile
__global__ __launch_bounds__(256) void dd_materialize_kernel(const float* __restrict__ A1, const float* __restrict__ A2,
                                                             const float* __restrict__ R1, bf16* __restrict__ Dm, bf16* __restrict__ Dmt,
                                                             const int dim, const int R, const float s) {
  __shared__ float coef[3][64];
  __shared__ bf16 tile[3][32][34];
  const int l = blockIdx.z, o0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
  if (threadIdx.x < 3 * R) {
    const int k = threadIdx.x / R, r = threadIdx.x - k * R;
    coef[k][r] = s * R1[r] * A1[(size_t)(3 * l + k) * R + r];
  }
  __syncthreads();
  bf16* dm = Dm + (size_t)l * 3 * dim * dim;
  bf16* dmt = Dmt + (size_t)l * 3 * dim * dim;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = threadIdx.x + 256 * j, el = idx >> 5, ol = idx & 31;
    const float* a = A2 + ((size_t)(e0 + el) * dim + o0 + ol) * R;
    float v[3] = {0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
      const float x = a[r];
      v[0] += coef[0][r] * x; v[1] += coef[1][r] * x; v[2] += coef[2][r] * x;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bf16 b = (bf16)v[k];
      dmt[(size_t)(e0 + el) * 3 * dim + k * dim + o0 + ol] = b;     // [e][k dim + o]: o fastest
      tile[k][el][ol] = b;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = threadIdx.x + 256 * j, ol = idx >> 5, el = idx & 31;
#pragma unroll
    for (int k = 0; k < 3; ++k) dm[(size_t)(k * dim + o0 + ol) * dim + e0 + el] = tile[k][el][ol];   // [k dim + o][e]: e fastest
  }
}

// dA2[p, :] = sum_lk c[lk][:] dD[lk][p]: one thread per row p of CP_A2
__global__ __launch_bounds__(256) void dd_grad_a2_kernel(const float* __restrict__ A1, const float* __restrict__ R1,
                                                         const float* __restrict__ dD, float* __restrict__ dA2, const size_t P,
                                                         const int nlk, const int R, const float s) {
  extern __shared__ float c[];   // [nlk][R]
  for (int i = threadIdx.x; i < nlk * R; i += 256) c[i] = s * R1[i % R] * A1[i];
  __syncthreads();
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float acc[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) acc[r] = 0.f;
  for (int lk = 0; lk < nlk; ++lk) {
    const float d = dD[(size_t)lk * P + p];
    const float* cl = c + lk * R;
#pragma unroll
    for (int r = 0; r < 64; ++r)
      if (r < R) acc[r] += cl[r] * d;
  }
  float* o = dA2 + p * R;
#pragma unroll
  for (int r = 0; r < 64; ++r)
    if (r < R) o[r] = acc[r];
}

// partial S[chunk][lk][r] = sum over the chunk's 256 rows p of dD[lk][p] A2[p, r]: thread -> one row (its A2 row in registers),
// wave sums, then the four waves in fixed order
__global__ __launch_bounds__(256) void dd_grad_s_kernel(const float* __restrict__ A2, const float* __restrict__ dD,
                                                        float* __restrict__ part, const size_t P, const int nlk, const int R) {
  __shared__ float red[4][64];   // [wave][r]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  float a[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) a[r] = (r < R && p < P) ? A2[p * R + r] : 0.f;
  for (int lk = 0; lk < nlk; ++lk) {
    const float d = p < P ? dD[(size_t)lk * P + p] : 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      if (r < R) {   // (wave-uniform)
        const float v = wave_sum(d * a[r]);
        if (lane == 0) red[wave][r] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < R)
      part[((size_t)blockIdx.x * nlk + lk) * R + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

// S[lk][r] = sum of the chunk partials (fixed order, four interleaved sums); dA1[lk][r] = s R1[r] S; dR1[r] = s sum_lk A1[lk][r] S
__global__ __launch_bounds__(256) void dd_grad_finish_kernel(const float* __restrict__ A1, const float* __restrict__ R1,
                                                             const float* __restrict__ part, float* __restrict__ dA1, float* __restrict__ dR1,
                                                             const int nchunk, const int nlk, const int R, const float s) {
  __shared__ float S[DD_MAXLK * 64];
  for (int i = threadIdx.x; i < nlk * R; i += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= nchunk; c += 4) {
      s0 += part[(size_t)c * nlk * R + i];
      s1 += part[(size_t)(c + 1) * nlk * R + i];
      s2 += part[(size_t)(c + 2) * nlk * R + i];
      s3 += part[(size_t)(c + 3) * nlk * R + i];
    }
    for (; c < nchunk; ++c) s0 += part[(size_t)c * nlk * R + i];
    const float v = (s0 + s1) + (s2 + s3);
    S[i] = v;
    dA1[i] = s * R1[i % R] * v;
  }
  __syncthreads();
  if (threadIdx.x < R) {
    float acc = 0.f;
    for (int lk = 0; lk < nlk; ++lk) acc += A1[lk * R + threadIdx.x] * S[lk * R + threadIdx.x];
    dR1[threadIdx.x] = s * acc;
  }
}

// out[j] = sum of `n` slabs (fixed order); the split-K partials of one dD
__global__ __launch_bounds__(256) void dd_sum_slabs_kernel(const float* __restrict__ slabs, const int n, const size_t stride, const size_t count,
                                                           float* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= count) return;
  float s0 = 0.f, s1 = 0.f;
  int c = 0;
  for (; c + 2 <= n; c += 2) {
    s0 += slabs[(size_t)c * stride + j];
    s1 += slabs[(size_t)(c + 1) * stride + j];
  }
  if (c < n) s0 += slabs[(size_t)c * stride + j];
  out[j] = s0 + s1;
}

// (dim % 128: the backward's dense dD = x^T dY comes from cara_gemm_tn_f32, whose output tiles are 128 x 128 -- refused here, at
// set-up, not in the middle of a backward pass)
bool dd_geom_ok(const cara_geom* g) {
  return g && g->cp_length == 2 && g->depth > 0 && 3 * g->depth <= DD_MAXLK && g->dim > 0 && g->dim % 128 == 0 && g->rank > 0 && g->rank <= 64;
}

}  // namespace

extern "C" int cara_dense_delta_materialize(const cara_geom* g, const cara_cp* cp, void* Dm, void* Dmt, void* stream) {
  if (!dd_geom_ok(g) || !cp || !cp->A1 || !cp->A2 || !cp->R1 || !Dm || !Dmt) return CARA_E_ARG;
  hipLaunchKernelGGL(dd_materialize_kernel, dim3(g->dim / 32, g->dim / 32, g->depth), dim3(256), 0, static_cast<hipStream_t>(stream), cp->A1,
                     cp->A2, cp->R1, static_cast<bf16*>(Dm), static_cast<bf16*>(Dmt), g->dim, g->rank, g->scale);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" size_t cara_dense_delta_grad_scratch_bytes(const cara_geom* g) {
  if (!dd_geom_ok(g)) return 0;
  const size_t P = (size_t)g->dim * g->dim, nchunk = (P + 255) / 256;
  return nchunk * 3 * g->depth * g->rank * sizeof(float);
}

extern "C" int cara_dense_delta_grad(const cara_geom* g, const cara_cp* cp, const float* dD, const cara_cp* grads, void* scratch, void* stream) {
  if (!dd_geom_ok(g) || !cp || !cp->A1 || !cp->A2 || !cp->R1 || !dD || !grads || !grads->A1 || !grads->A2 || !grads->R1 || !scratch)
    return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t P = (size_t)g->dim * g->dim;
  const int nlk = 3 * g->depth, R = g->rank;
  const int nchunk = (int)((P + 255) / 256);
  float* part = static_cast<float*>(scratch);
  hipLaunchKernelGGL(dd_grad_a2_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), (size_t)nlk * R * sizeof(float), st, cp->A1, cp->R1, dD, grads->A2, P,
                     nlk, R, g->scale);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(dd_grad_s_kernel, dim3(nchunk), dim3(256), 0, st, cp->A2, dD, part, P, nlk, R);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(dd_grad_finish_kernel, dim3(1), dim3(256), 0, st, cp->A1, cp->R1, part, grads->A1, grads->R1, nchunk, nlk, R, g->scale);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_sum_slabs_f32(const float* slabs, int nslab, size_t slab_stride, size_t count, float* out, void* stream) {
  if (!slabs || !out || nslab <= 0 || count == 0) return CARA_E_ARG;
  hipLaunchKernelGGL(dd_sum_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), slabs, nslab,
                     slab_stride, count, out);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
