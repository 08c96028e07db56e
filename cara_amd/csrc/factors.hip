// CP factor preparation and gradient scatter (gfx950).  O((in+out) * R) work per linear:
// everything the reference does by materialising dW with tensorly.cp_to_tensor
// (/root/reference/src/cara/cara.py:26-34,51-56,76-80,88-91) collapses, in factored form, to
// building per layer  U [in,Rp]  and  Vs = s * g (.) V [out,Rp]   and, in
// backward, to scattering dU / dVs / dc back onto the 12 shared tensors (A.4).
#include "common.h"

namespace {

struct PackDims {
  int depth, dim, heads, hd, rank, Rp;
  float s;
};

// logical matrices of one layer, in pack order
enum { M_U_QKV, M_VS_QKV, M_U_PROJ, M_VS_PROJ, M_U_FC1, M_VS_FC1, M_U_FC2, M_VS_FC2, M_COUNT };

__host__ __device__ inline int mat_rows(int m, int dim) {
  switch (m) {
    case M_VS_QKV: return 3 * dim;
    case M_VS_FC1: case M_U_FC2: return 4 * dim;
    default: return dim;
  }
}

__device__ __forceinline__ float factor_value(const PackDims& g, const cara_cp& cp, int l, int m, int row, int r) {
  if (r >= g.rank) return 0.f;
  const int R = g.rank, dim = g.dim;
  switch (m) {
    case M_U_QKV: return cp.A2[row * R + r];
    case M_VS_QKV: {
      const int k = row / dim, c = row - k * dim, hh = c / g.hd, d = c - hh * g.hd;
      return g.s * cp.R1[r] * cp.A1[(3 * l + k) * R + r] * cp.A3[hh * R + r] * cp.A4[d * R + r];
    }
    case M_U_PROJ: case M_U_FC1: return cp.P3[row * R + r];
    case M_VS_PROJ: return g.s * cp.R2[r] * cp.P1[(9 * l) * R + r] * cp.P2[row * R + r];
    case M_VS_FC1: {
      const int a = row / dim, j = row - a * dim;
      return g.s * cp.R2[r] * cp.P1[(9 * l + 1 + a) * R + r] * cp.P2[j * R + r];
    }
    case M_U_FC2: {
      const int a = row / dim, j = row - a * dim;
      return cp.P1[(9 * l + 5 + a) * R + r] * cp.P2[j * R + r];
    }
    default: return g.s * cp.R2[r] * cp.P3[row * R + r];  // M_VS_FC2
  }
}

struct PackOffsets {  // byte offsets inside one layer
  size_t rm[M_COUNT];   // row-major [rows, Rp]
  size_t tr[M_COUNT];   // transposed [Rp, rows]
  size_t bias[3];
  size_t layer_stride;
};

__host__ inline PackOffsets make_offsets(int dim, int Rp) {
  PackOffsets o;
  size_t off = 0;
  for (int m = 0; m < M_COUNT; ++m) {
    const size_t bytes = (size_t)mat_rows(m, dim) * Rp * 2;
    o.tr[m] = off; off += bytes;
    o.rm[m] = off; off += bytes;
  }
  o.bias[0] = off; off += (size_t)dim * 4;
  o.bias[1] = off; off += (size_t)4 * dim * 4;
  o.bias[2] = off; off += (size_t)dim * 4;
  o.layer_stride = (off + 255) & ~(size_t)255;
  return o;
}

// grid.y = layer * M_COUNT + matrix; grid.x covers 2 * rows * Rp elements: first half writes the
// row-major copy (r fastest), second half the transposed copy (row fastest) -> both coalesced.
__global__ __launch_bounds__(256) void prep_kernel(PackDims g, cara_cp cp, PackOffsets po, char* __restrict__ pack) {
  const int l = blockIdx.y / M_COUNT, m = blockIdx.y - l * M_COUNT;
  const int rows = mat_rows(m, g.dim);
  const int n = rows * g.Rp;
  int e = blockIdx.x * 256 + threadIdx.x;
  char* base = pack + (size_t)l * po.layer_stride;
  if (e < n) {
    const int row = e / g.Rp, r = e - row * g.Rp;
    reinterpret_cast<bf16*>(base + po.rm[m])[e] = (bf16)factor_value(g, cp, l, m, row, r);
  } else if (e < 2 * n) {
    e -= n;
    const int r = e / rows, row = e - r * rows;
    reinterpret_cast<bf16*>(base + po.tr[m])[e] = (bf16)factor_value(g, cp, l, m, row, r);
  }
}

__global__ __launch_bounds__(256) void prep_bias_kernel(PackDims g, cara_cp cp, PackOffsets po,
                                                        const float* __restrict__ bp, const float* __restrict__ b1,
                                                        const float* __restrict__ b2, char* __restrict__ pack) {
  const int l = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int dim = g.dim;
  char* base = pack + (size_t)l * po.layer_stride;
  if (e < dim) {
    reinterpret_cast<float*>(base + po.bias[0])[e] = bp[l * dim + e] + g.s * cp.bias1[e];
    reinterpret_cast<float*>(base + po.bias[2])[e] = b2[l * dim + e] + g.s * cp.bias3[e];
  }
  if (e < 4 * dim) reinterpret_cast<float*>(base + po.bias[1])[e] = b1[l * 4 * dim + e] + g.s * cp.bias2[e];
}

// ---------------------------------------------------------------------------------------------
// gradient scatter (A.4).  All layer inputs are fp32 [depth, rows, Rp]; only r < rank is read.
// ---------------------------------------------------------------------------------------------

// (a) outputs indexed (j, r) that sum over layers: dA2, dP3, dP2, bias grads
__global__ __launch_bounds__(256) void grad_rowwise_kernel(PackDims g, cara_cp cp, cara_layer_grads lg, cara_cp out) {
  const int R = g.rank, Rp = g.Rp, dim = g.dim, L = g.depth;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < dim * R) {
    const int j = e / R, r = e - j * R;
    float a2 = 0.f, p3 = 0.f, p2 = 0.f;
    const float sr2 = g.s * cp.R2[r];
    for (int l = 0; l < L; ++l) {
      a2 += lg.dU_qkv[((size_t)l * dim + j) * Rp + r];
      p3 += lg.dU_proj[((size_t)l * dim + j) * Rp + r] + lg.dU_fc1[((size_t)l * dim + j) * Rp + r] +
            sr2 * lg.dVs_fc2[((size_t)l * dim + j) * Rp + r];
      p2 += sr2 * cp.P1[(9 * l) * R + r] * lg.dVs_proj[((size_t)l * dim + j) * Rp + r];
      for (int a = 0; a < 4; ++a) {
        p2 += sr2 * cp.P1[(9 * l + 1 + a) * R + r] * lg.dVs_fc1[((size_t)l * 4 * dim + a * dim + j) * Rp + r];
        p2 += cp.P1[(9 * l + 5 + a) * R + r] * lg.dU_fc2[((size_t)l * 4 * dim + a * dim + j) * Rp + r];
      }
    }
    out.A2[e] = a2;
    out.P3[e] = p3;
    out.P2[e] = p2;
  }
  if (e < 4 * dim) {
    float b = 0.f;
    for (int l = 0; l < L; ++l) b += lg.dc_fc1[(size_t)l * 4 * dim + e];
    out.bias2[e] = g.s * b;
  }
  if (e < dim) {
    float b1 = 0.f, b3 = 0.f;
    for (int l = 0; l < L; ++l) {
      b1 += lg.dc_proj[(size_t)l * dim + e];
      b3 += lg.dc_fc2[(size_t)l * dim + e];
    }
    out.bias1[e] = g.s * b1;
    out.bias3[e] = g.s * b3;
  }
}

// (b) column reductions Z[l, slot, r] = sum_rows W[row, r] * F[row, r], written (unscaled by the
// lambda) into the dA1 / dP1 rows; slots per layer: 0-2 qkv(k), 3 proj, 4-7 fc1(a), 8-11 fc2(a)
__global__ __launch_bounds__(256) void grad_colred_kernel(PackDims g, cara_cp cp, cara_layer_grads lg, cara_cp out) {
  __shared__ float red[256];
  const int R = g.rank, Rp = g.Rp, dim = g.dim;
  const int l = blockIdx.x / 12, slot = blockIdx.x - l * 12;
  const int r = threadIdx.x % 32, part = threadIdx.x / 32;  // 8 row partitions; R <= 64 -> loop over r groups
  for (int rb = 0; rb < R; rb += 32) {
    const int rr = rb + r;
    float z = 0.f;
    if (rr < R) {
      for (int row = part; row < dim; row += 8) {
        float w, f;
        if (slot < 3) {
          const int hh = row / g.hd, d = row - hh * g.hd;
          w = lg.dVs_qkv[((size_t)l * 3 * dim + slot * dim + row) * Rp + rr];
          f = g.s * cp.A3[hh * R + rr] * cp.A4[d * R + rr];
        } else if (slot == 3) {
          w = lg.dVs_proj[((size_t)l * dim + row) * Rp + rr];
          f = g.s * cp.P2[row * R + rr];
        } else if (slot < 8) {
          w = lg.dVs_fc1[((size_t)l * 4 * dim + (slot - 4) * dim + row) * Rp + rr];
          f = g.s * cp.P2[row * R + rr];
        } else {
          w = lg.dU_fc2[((size_t)l * 4 * dim + (slot - 8) * dim + row) * Rp + rr];
          f = cp.P2[row * R + rr];
        }
        z += w * f;
      }
    }
    red[threadIdx.x] = z;
    __syncthreads();
    if (part == 0 && rr < R) {
      float s = 0.f;
      for (int p = 0; p < 8; ++p) s += red[p * 32 + r];
      if (slot < 3) out.A1[(3 * l + slot) * R + rr] = s;
      else if (slot == 3) out.P1[(9 * l) * R + rr] = s;
      else out.P1[(9 * l + 1 + (slot - 4)) * R + rr] = s;   // 4..7 -> 9l+1..4 ; 8..11 -> 9l+5..8
    }
    __syncthreads();
  }
}

// (c) dA3[hh, r] and dA4[d, r]: one block per (layer, k) writes its partial (already multiplied by
// A1[3l+k, r]) into scratch; the final kernel sums the 3*depth partials in a fixed order.
//   scratch layout: pa3 [3L][H][R], pa4 [3L][hd][R], zv [L][R]
__global__ __launch_bounds__(256) void grad_a34_partial_kernel(PackDims g, cara_cp cp, cara_layer_grads lg,
                                                               float* __restrict__ scratch) {
  const int R = g.rank, Rp = g.Rp, dim = g.dim, H = g.heads, hd = g.hd, L = g.depth;
  const int lk = blockIdx.x, l = lk / 3, k = lk - 3 * l;
  const float* W = lg.dVs_qkv + ((size_t)l * 3 * dim + (size_t)k * dim) * Rp;   // [H*hd, Rp]
  float* pa3 = scratch + (size_t)lk * H * R;
  float* pa4 = scratch + (size_t)3 * L * H * R + (size_t)lk * hd * R;
  for (int e = threadIdx.x; e < H * R; e += 256) {
    const int hh = e / R, r = e - hh * R;
    float in = 0.f;
    for (int d = 0; d < hd; ++d) in += W[(size_t)(hh * hd + d) * Rp + r] * cp.A4[d * R + r];
    pa3[e] = cp.A1[lk * R + r] * in;
  }
  for (int e = threadIdx.x; e < hd * R; e += 256) {
    const int d = e / R, r = e - d * R;
    float in = 0.f;
    for (int hh = 0; hh < H; ++hh) in += W[(size_t)(hh * hd + d) * Rp + r] * cp.A3[hh * R + r];
    pa4[e] = cp.A1[lk * R + r] * in;
  }
}

// zv[l, r] = sum_c dVs_fc2[l, c, r] * P3[c, r]   (the R2-gradient carried by fc2's output factor)
__global__ __launch_bounds__(256) void grad_zv_kernel(PackDims g, cara_cp cp, cara_layer_grads lg, float* __restrict__ scratch) {
  __shared__ float red[256];
  const int R = g.rank, Rp = g.Rp, dim = g.dim, H = g.heads, hd = g.hd, L = g.depth;
  float* zv = scratch + (size_t)3 * L * (H + hd) * R;
  const int l = blockIdx.x;
  const int r = threadIdx.x % 32, part = threadIdx.x / 32;
  for (int rb = 0; rb < R; rb += 32) {
    const int rr = rb + r;
    float v = 0.f;
    if (rr < R)
      for (int c = part; c < dim; c += 8) v += lg.dVs_fc2[((size_t)l * dim + c) * Rp + rr] * cp.P3[c * R + rr];
    red[threadIdx.x] = v;
    __syncthreads();
    if (part == 0 && rr < R) {
      float s = 0.f;
      for (int p = 0; p < 8; ++p) s += red[p * 32 + r];
      zv[l * R + rr] = s;
    }
    __syncthreads();
  }
}

// (d) finish: dA3/dA4 from the partials, lambda gradients, then scale the Z rows by the lambda.
//   dR1 = sum A1 (.) Z ; dR2 = sum over the R2-carrying P1 rows (9l .. 9l+4) of P1 (.) Z + s * sum_l zv
__global__ __launch_bounds__(256) void grad_finish_kernel(PackDims g, cara_cp cp, const float* __restrict__ scratch, cara_cp out) {
  const int R = g.rank, H = g.heads, hd = g.hd, L = g.depth;
  const float* pa3 = scratch;
  const float* pa4 = scratch + (size_t)3 * L * H * R;
  const float* zv = scratch + (size_t)3 * L * (H + hd) * R;
  for (int e = threadIdx.x; e < (H + hd) * R; e += 256) {
    const bool is3 = e < H * R;
    const int e2 = is3 ? e : e - H * R;
    const int r = e2 % R;
    const int per = is3 ? H * R : hd * R;
    const float* p = is3 ? pa3 : pa4;
    float acc = 0.f;
    for (int lk = 0; lk < 3 * L; ++lk) acc += p[(size_t)lk * per + e2];
    (is3 ? out.A3 : out.A4)[e2] = g.s * cp.R1[r] * acc;
  }
  for (int rr = threadIdx.x; rr < R; rr += 256) {
    float d1 = 0.f, d2 = 0.f;
    for (int l = 0; l < L; ++l) d2 += zv[l * R + rr];
    d2 *= g.s;
    for (int row = 0; row < 3 * L; ++row) {
      const float z = out.A1[row * R + rr];
      d1 += cp.A1[row * R + rr] * z;
      out.A1[row * R + rr] = cp.R1[rr] * z;
    }
    for (int row = 0; row < 9 * L; ++row) {
      if (row % 9 < 5) {
        const float z = out.P1[row * R + rr];
        d2 += cp.P1[row * R + rr] * z;
        out.P1[row * R + rr] = cp.R2[rr] * z;
      }
    }
    out.R1[rr] = d1;
    out.R2[rr] = d2;
  }
}

PackDims dims_of(const cara_geom* g) {
  PackDims d;
  d.depth = g->depth; d.dim = g->dim; d.heads = g->heads; d.hd = g->dim / g->heads;
  d.rank = g->rank; d.Rp = g->Rp; d.s = g->scale;
  return d;
}
bool geom_ok(const cara_geom* g) {
  return g && g->depth > 0 && g->dim > 0 && g->heads > 0 && g->dim % g->heads == 0 && g->rank > 0 &&
         g->rank <= g->Rp && (g->Rp == 32 || g->Rp == 64);
}
bool cp_ok(const cara_cp* c) {
  return c && c->A1 && c->A2 && c->A3 && c->A4 && c->P1 && c->P2 && c->P3 && c->R1 && c->R2 && c->bias1 && c->bias2 && c->bias3;
}

}  // namespace

extern "C" int cara_pack_offsets(const cara_geom* g, cara_pack_layout* out) {
  if (!geom_ok(g) || !out) return CARA_E_ARG;
  const PackOffsets o = make_offsets(g->dim, g->Rp);
  out->Ut_qkv = o.tr[M_U_QKV]; out->U_qkv = o.rm[M_U_QKV]; out->Vs_qkv = o.rm[M_VS_QKV]; out->Vst_qkv = o.tr[M_VS_QKV];
  out->Ut_proj = o.tr[M_U_PROJ]; out->U_proj = o.rm[M_U_PROJ]; out->Vs_proj = o.rm[M_VS_PROJ]; out->Vst_proj = o.tr[M_VS_PROJ];
  out->Ut_fc1 = o.tr[M_U_FC1]; out->U_fc1 = o.rm[M_U_FC1]; out->Vs_fc1 = o.rm[M_VS_FC1]; out->Vst_fc1 = o.tr[M_VS_FC1];
  out->Ut_fc2 = o.tr[M_U_FC2]; out->U_fc2 = o.rm[M_U_FC2]; out->Vs_fc2 = o.rm[M_VS_FC2]; out->Vst_fc2 = o.tr[M_VS_FC2];
  out->bias_proj = o.bias[0]; out->bias_fc1 = o.bias[1]; out->bias_fc2 = o.bias[2];
  out->layer_stride = o.layer_stride;
  out->total = o.layer_stride * g->depth;
  return CARA_OK;
}

extern "C" int cara_factor_prep(const cara_geom* g, const cara_cp* cp, const float* base_bias_proj,
                                const float* base_bias_fc1, const float* base_bias_fc2, void* pack, void* stream) {
  if (!geom_ok(g) || !cp_ok(cp) || !base_bias_proj || !base_bias_fc1 || !base_bias_fc2 || !pack) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackDims d = dims_of(g);
  const PackOffsets po = make_offsets(g->dim, g->Rp);
  const int maxn = 2 * 4 * g->dim * g->Rp;
  hipLaunchKernelGGL(prep_kernel, dim3((maxn + 255) / 256, g->depth * M_COUNT), dim3(256), 0, st, d, *cp, po, (char*)pack);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(prep_bias_kernel, dim3((4 * g->dim + 255) / 256, g->depth), dim3(256), 0, st, d, *cp, po,
                     base_bias_proj, base_bias_fc1, base_bias_fc2, (char*)pack);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" size_t cara_factor_grad_scratch_bytes(const cara_geom* g) {
  if (!geom_ok(g)) return 0;
  const size_t L = g->depth, R = g->rank, H = g->heads, hd = g->dim / g->heads;
  return (3 * L * (H + hd) * R + L * R) * sizeof(float);
}

extern "C" int cara_factor_grad_reduce(const cara_geom* g, const cara_cp* cp, const cara_layer_grads* lg,
                                       const cara_cp* grads, void* scratch, void* stream) {
  if (!geom_ok(g) || !cp_ok(cp) || !cp_ok(grads) || !lg || !scratch) return CARA_E_ARG;
  if (!lg->dU_qkv || !lg->dVs_qkv || !lg->dU_proj || !lg->dVs_proj || !lg->dU_fc1 || !lg->dVs_fc1 || !lg->dU_fc2 ||
      !lg->dVs_fc2 || !lg->dc_proj || !lg->dc_fc1 || !lg->dc_fc2)
    return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackDims d = dims_of(g);
  const int n1 = g->dim * g->rank > 4 * g->dim ? g->dim * g->rank : 4 * g->dim;
  hipLaunchKernelGGL(grad_rowwise_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, d, *cp, *lg, *grads);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(grad_colred_kernel, dim3(g->depth * 12), dim3(256), 0, st, d, *cp, *lg, *grads);
  CARA_CHECK_LAUNCH();
  float* sc = static_cast<float*>(scratch);
  hipLaunchKernelGGL(grad_a34_partial_kernel, dim3(3 * g->depth), dim3(256), 0, st, d, *cp, *lg, sc);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(grad_zv_kernel, dim3(g->depth), dim3(256), 0, st, d, *cp, *lg, sc);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(grad_finish_kernel, dim3(1), dim3(256), 0, st, d, *cp, sc, *grads);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
