// CP factor preparation and gradient scatter (gfx950).  O((in+out) * R) work per linear:
// everything the reference does by materialising dW with tensorly.cp_to_tensor
// (/root/reference/src/cara/cara.py:26-34,51-56,76-80,88-91) collapses, in factored form, to
// building per layer  U [in,Rp]  and  Vs = s * g (.) V [out,Rp]   and, in
// backward, to scattering dU / dVs / dc back onto the 12 shared tensors (A.4).
#include <stdlib.h>

#include "common.h"

namespace {

struct PackDims {
  int depth, dim, heads, hd, rank, Rp;
  float s;
  int cpl;   // order of the QKV tensorisation: 2, 3, 4 or 5 (cara_geom::cp_length)
  // loss scaling (cara_factor_grad_reduce_ex): every FINAL gradient is written through gout() -- times 1 / *loss_scale, and
  // *found_inf raised when the value is not finite (all writers store the same 1.0f: a benign race)
  const float* loss_scale;
  float* found_inf;
};
struct GOut {
  float gs;
  float* found;
  __device__ __forceinline__ float operator()(float v) const {
    const float r = v * gs;
    if (found && !(fabsf(r) <= 3.0e38f)) *found = 1.f;
    return r;
  }
};
__device__ __forceinline__ GOut gout_of(const PackDims& g) { return GOut{g.loss_scale ? 1.f / *g.loss_scale : 1.f, g.found_inf}; }

// The QKV adapter of block l, projection k, in factored form for every supported order (cara_geom::cp_length):
//   dW_k[e, o] = sum_r  R1[r] * qkv_coef(l, k, r) * qkv_in(e, r) * qkv_out(o, r)
__device__ __forceinline__ float qkv_coef(const PackDims& g, const cara_cp& cp, int l, int k, int r) {
  return g.cpl == 5 ? cp.A1[l * g.rank + r] * cp.A2[k * g.rank + r] : cp.A1[(3 * l + k) * g.rank + r];
}
__device__ __forceinline__ float qkv_in(const PackDims& g, const cara_cp& cp, int e, int r) {
  return (g.cpl == 5 ? cp.A3 : cp.A2)[e * g.rank + r];
}
__device__ __forceinline__ float qkv_out(const PackDims& g, const cara_cp& cp, int o, int r) {
  if (g.cpl == 3) return cp.A3[o * g.rank + r];
  const int hh = o / g.hd, d = o - hh * g.hd;
  return g.cpl == 5 ? cp.A4[hh * g.rank + r] * cp.A5[d * g.rank + r] : cp.A3[hh * g.rank + r] * cp.A4[d * g.rank + r];
}

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
  if (g.cpl == 2 && (m == M_U_QKV || m == M_VS_QKV)) return 0.f;   // order 2: the QKV adapter is not low-rank (dense_delta.hip)
  const int R = g.rank, dim = g.dim;
  switch (m) {
    case M_U_QKV: return qkv_in(g, cp, row, r);
    case M_VS_QKV: {
      const int k = row / dim, c = row - k * dim;
      return g.s * cp.R1[r] * qkv_coef(g, cp, l, k, r) * qkv_out(g, cp, c, r);
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

// grid.y = layer * M_COUNT + matrix; grid.x covers 2 * rows * Rp / 8 threads of EIGHT elements each: the first half writes the
// row-major copy (8 consecutive r of a row), the second half the transposed copy (8 consecutive rows of an r) -> one 16-byte
// store per thread, both coalesced.  (One element per thread: 30 us per step for 19 MB of output.)
__global__ __launch_bounds__(256) void prep_kernel(PackDims g, cara_cp cp, PackOffsets po, char* __restrict__ pack) {
  const int l = blockIdx.y / M_COUNT, m = blockIdx.y - l * M_COUNT;
  const int rows = mat_rows(m, g.dim);
  const int n8 = rows * g.Rp / 8;          // (rows and Rp are multiples of 8)
  int e = blockIdx.x * 256 + threadIdx.x;
  char* base = pack + (size_t)l * po.layer_stride;
  bf16x8 v;
  if (e < n8) {
    const int rp8 = g.Rp / 8;
    const int row = e / rp8, r0 = (e - row * rp8) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)factor_value(g, cp, l, m, row, r0 + j);
    reinterpret_cast<bf16x8*>(base + po.rm[m])[e] = v;
  } else if (e < 2 * n8) {
    e -= n8;
    const int rows8 = rows / 8;
    const int r = e / rows8, row0 = (e - r * rows8) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)factor_value(g, cp, l, m, row0 + j, r);
    reinterpret_cast<bf16x8*>(base + po.tr[m])[e] = v;
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

// Two launches.  Stage 1 runs the four independent reductions side by side in ONE grid (block ranges), with the
// long row sums cut into GS_SPLIT partials so that no thread walks more than ~24 rows; stage 2 sums the partials in
// a fixed order (bitwise reproducible, no float atomics).  As five back-to-back launches of 36..144 blocks whose
// threads walked 96..150 rows each, the same arithmetic took 210 us per step -- all of it latency.
//   scratch: pa3 [3L][H][R], pa4 [3L][hd][R], zvp [L][GS_SPLIT][R], zp [12L][GS_SPLIT][R]
constexpr int GS_SPLIT = 8;   // (r05: 4 -> 8, six rows per thread of the column reductions instead of twelve)
struct GradScratch {
  float *pa3, *pa4, *zvp, *zp;
};
__host__ __device__ inline GradScratch grad_scratch(float* sc, int L, int H, int hd, int R) {
  GradScratch g;
  g.pa3 = sc;
  g.pa4 = g.pa3 + (size_t)3 * L * H * R;
  g.zvp = g.pa4 + (size_t)3 * L * hd * R;
  g.zp = g.zvp + (size_t)L * GS_SPLIT * R;
  return g;
}

// (a) outputs indexed (j, r) that sum over layers: dA2, dP3, dP2, bias grads.  A block = 64 outputs x FOUR layer groups (a thread
// walks depth / 4 layers: one batch of independent loads instead of a chain of twelve layers' -- the range was 15 us of the launch);
// the four partial sums meet in LDS and are added in a fixed order.
constexpr int ROW_E = 64;
__device__ __forceinline__ void grad_rowwise(const PackDims& g, const cara_cp& cp, const cara_layer_grads& lg, const cara_cp& out,
                                             const int blk, float (*red)[4][ROW_E]) {
  const int R = g.rank, Rp = g.Rp, dim = g.dim, L = g.depth;
  const GOut W = gout_of(g);
  const int el = threadIdx.x % ROW_E, grp = threadIdx.x / ROW_E;     // grp 0 .. 3
  const int e = blk * ROW_E + el;
  const int l0 = L * grp / 4, l1 = L * (grp + 1) / 4;
  float a2 = 0.f, p3 = 0.f, p2 = 0.f, a3o = 0.f;
  int r = 0;
  if (e < dim * R) {
    const int j = e / R;
    r = e - j * R;
    const float sr2 = g.s * cp.R2[r];
#pragma unroll 3
    for (int l = l0; l < l1; ++l) {
      if (g.cpl != 2) a2 += lg.dU_qkv[((size_t)l * dim + j) * Rp + r];
      if (g.cpl == 3)   // order 3: the out factor A3 [dim, R] is a plain row-wise sum over blocks and projections
        for (int k = 0; k < 3; ++k) a3o += cp.A1[(3 * l + k) * R + r] * lg.dVs_qkv[((size_t)l * 3 * dim + k * dim + j) * Rp + r];
      p3 += lg.dU_proj[((size_t)l * dim + j) * Rp + r] + lg.dU_fc1[((size_t)l * dim + j) * Rp + r] +
            sr2 * lg.dVs_fc2[((size_t)l * dim + j) * Rp + r];
      p2 += sr2 * cp.P1[(9 * l) * R + r] * lg.dVs_proj[((size_t)l * dim + j) * Rp + r];
      for (int a = 0; a < 4; ++a) {
        p2 += sr2 * cp.P1[(9 * l + 1 + a) * R + r] * lg.dVs_fc1[((size_t)l * 4 * dim + a * dim + j) * Rp + r];
        p2 += cp.P1[(9 * l + 5 + a) * R + r] * lg.dU_fc2[((size_t)l * 4 * dim + a * dim + j) * Rp + r];
      }
    }
  }
  float b2 = 0.f, b1 = 0.f, b3 = 0.f;
  if (e < 4 * dim)
    for (int l = l0; l < l1; ++l) b2 += lg.dc_fc1[(size_t)l * 4 * dim + e];
  if (e < dim)
    for (int l = l0; l < l1; ++l) {
      b1 += lg.dc_proj[(size_t)l * dim + e];
      b3 += lg.dc_fc2[(size_t)l * dim + e];
    }
  auto total = [&](const float v, const int slot) {   // (block-uniform call sequence: every thread takes part in every barrier)
    __syncthreads();
    red[slot & 3][grp][el] = v;
    __syncthreads();
    return (red[slot & 3][0][el] + red[slot & 3][1][el]) + (red[slot & 3][2][el] + red[slot & 3][3][el]);
  };
  const float ta2 = total(a2, 0), tp3 = total(p3, 1), tp2 = total(p2, 2), ta3 = total(a3o, 3);
  const float tb2 = total(b2, 0), tb1 = total(b1, 1), tb3 = total(b3, 2);
  if (grp != 0) return;
  if (e < dim * R) {
    if (g.cpl != 2) (g.cpl == 5 ? out.A3 : out.A2)[e] = W(ta2);   // gradient of the in factor (order 2: cara_dense_delta_grad)
    if (g.cpl == 3) out.A3[e] = W(g.s * cp.R1[r] * ta3);
    out.P3[e] = W(tp3);
    out.P2[e] = W(tp2);
  }
  if (e < 4 * dim) out.bias2[e] = W(g.s * tb2);
  if (e < dim) {
    out.bias1[e] = W(g.s * tb1);
    out.bias3[e] = W(g.s * tb3);
  }
}

// (b) column reductions Z[l, slot, r] = sum_rows W[row, r] * F[row, r] (slots per layer: 0-2 qkv(k), 3 proj,
// 4-7 fc1(a), 8-11 fc2(a)), and (b') zv[l, r] = sum_c dVs_fc2[l, c, r] * P3[c, r] as slot 12: one block per
// (layer, slot, split) sums its quarter of the rows -> zp / zvp partials
__device__ __forceinline__ void grad_colred_part(const PackDims& g, const cara_cp& cp, const cara_layer_grads& lg, const GradScratch& sc,
                                                 const int l, const int slot, const int split, float* red) {
  const int R = g.rank, Rp = g.Rp, dim = g.dim;
  // Rr lanes of r (16 at rank <= 16: no idle half) x 256 / Rr row partitions; R <= 64 -> loop over r groups
  const int Rr = R <= 16 ? 16 : 32, NP = 256 / Rr;
  const int r = threadIdx.x % Rr, part = threadIdx.x / Rr;
  const int row0 = (int)((long)dim * split / GS_SPLIT), row1 = (int)((long)dim * (split + 1) / GS_SPLIT);
  for (int rb = 0; rb < R; rb += Rr) {
    const int rr = rb + r;
    float z = 0.f;
    if (rr < R) {
#pragma unroll 6
      for (int row = row0 + part; row < row1; row += NP) {
        float w, f;
        if (slot < 3) {
          if (g.cpl == 2) break;   // (no factored QKV adapter)
          w = lg.dVs_qkv[((size_t)l * 3 * dim + slot * dim + row) * Rp + rr];
          f = g.s * qkv_out(g, cp, row, rr);
        } else if (slot == 3) {
          w = lg.dVs_proj[((size_t)l * dim + row) * Rp + rr];
          f = g.s * cp.P2[row * R + rr];
        } else if (slot < 8) {
          w = lg.dVs_fc1[((size_t)l * 4 * dim + (slot - 4) * dim + row) * Rp + rr];
          f = g.s * cp.P2[row * R + rr];
        } else if (slot < 12) {
          w = lg.dU_fc2[((size_t)l * 4 * dim + (slot - 8) * dim + row) * Rp + rr];
          f = cp.P2[row * R + rr];
        } else {
          w = lg.dVs_fc2[((size_t)l * dim + row) * Rp + rr];
          f = cp.P3[row * R + rr];
        }
        z += w * f;
      }
    }
    red[threadIdx.x] = z;
    __syncthreads();
    if (part == 0 && rr < R) {
      float s = 0.f;
      for (int p = 0; p < NP; ++p) s += red[p * Rr + r];
      if (slot < 12) sc.zp[((size_t)(l * 12 + slot) * GS_SPLIT + split) * R + rr] = s;
      else sc.zvp[((size_t)l * GS_SPLIT + split) * R + rr] = s;
    }
    __syncthreads();
  }
}

// (c) dA3[hh, r] and dA4[d, r]: partials per (layer, k), already multiplied by A1[3l+k, r]; piece 0 of a (layer, k)
// does pa3, pieces 1..4 a quarter of pa4 each
__device__ __forceinline__ void grad_a34_part(const PackDims& g, const cara_cp& cp, const cara_layer_grads& lg, const GradScratch& sc,
                                              const int lk, const int piece) {
  const int R = g.rank, Rp = g.Rp, dim = g.dim, H = g.heads, hd = g.hd;
  const int l = lk / 3, k = lk - 3 * l;
  if (g.cpl == 3 || g.cpl == 2) return;   // no head / head-dim factors (order 3: grad_rowwise did A3)
  const float* W = lg.dVs_qkv + ((size_t)l * 3 * dim + (size_t)k * dim) * Rp;   // [H*hd, Rp]
  const float* Fh = g.cpl == 5 ? cp.A4 : cp.A3;   // head factor [H, R]
  const float* Fd = g.cpl == 5 ? cp.A5 : cp.A4;   // head-dim factor [hd, R]
  if (piece == 0) {
    float* pa3 = sc.pa3 + (size_t)lk * H * R;
    for (int e = threadIdx.x; e < H * R; e += 256) {
      const int hh = e / R, r = e - hh * R;
      float in = 0.f;
#pragma unroll 16
      for (int d = 0; d < hd; ++d) in += W[(size_t)(hh * hd + d) * Rp + r] * Fd[d * R + r];
      pa3[e] = qkv_coef(g, cp, l, k, r) * in;
    }
  } else {
    float* pa4 = sc.pa4 + (size_t)lk * hd * R;
    const int n = hd * R, e0 = (int)((long)n * (piece - 1) / 4), e1 = (int)((long)n * piece / 4);
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
      const int d = e / R, r = e - d * R;
      float in = 0.f;
#pragma unroll 12
      for (int hh = 0; hh < H; ++hh) in += W[(size_t)(hh * hd + d) * Rp + r] * Fh[hh * R + r];
      pa4[e] = qkv_coef(g, cp, l, k, r) * in;
    }
  }
}

// stage 1: block ranges [rowwise | colred + zv: L * 13 * GS_SPLIT | a34: 3L * 5]
__global__ __launch_bounds__(256) void grad_stage1_kernel(PackDims g, cara_cp cp, cara_layer_grads lg, cara_cp out, float* __restrict__ scratch,
                                                          int nb_row, int b0) {
  __shared__ float red_all[4 * 4 * ROW_E];
  float* red = red_all;
  const GradScratch sc = grad_scratch(scratch, g.depth, g.heads, g.hd, g.rank);
  int b = blockIdx.x + b0;   // (b0: CARA_GRAD_STAGE1_SPLIT=1 launches the three block ranges one after the other, to time them)
  if (b < nb_row) {
    grad_rowwise(g, cp, lg, out, b, reinterpret_cast<float (*)[4][ROW_E]>(red_all));
    return;
  }
  b -= nb_row;
  const int nb_col = g.depth * 13 * GS_SPLIT;
  if (b < nb_col) {
    const int split = b % GS_SPLIT, ls = b / GS_SPLIT;
    grad_colred_part(g, cp, lg, sc, ls / 13, ls % 13, split, red);
    return;
  }
  b -= nb_col;
  grad_a34_part(g, cp, lg, sc, b / 5, b % 5);
}

// stage 2.  Blocks [0, nb_a): dA3 / dA4 = s R1 (.) sum of the 3L partials.  Last block: Z = sum of the splits,
// the lambda gradients  dR1 = sum A1 (.) Z ;  dR2 = sum over the R2-carrying P1 rows (9l .. 9l+4) of P1 (.) Z
// + s * sum_l zv ,  and the Z rows scaled by their lambda into dA1 / dP1.
__global__ __launch_bounds__(256) void grad_stage2_kernel(PackDims g, cara_cp cp, float* __restrict__ scratch, cara_cp out, int nb_a) {
  const int R = g.rank, H = g.heads, hd = g.hd, L = g.depth;
  const GradScratch sc = grad_scratch(scratch, L, H, hd, R);
  const GOut W = gout_of(g);
  if ((int)blockIdx.x < nb_a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (g.cpl != 3 && g.cpl != 2 && e < (H + hd) * R) {
      const bool is3 = e < H * R;
      const int e2 = is3 ? e : e - H * R;
      const int r = e2 % R;
      const int per = is3 ? H * R : hd * R;
      const float* p = is3 ? sc.pa3 : sc.pa4;
      float acc = 0.f;
#pragma unroll 12
      for (int lk = 0; lk < 3 * L; ++lk) acc += p[(size_t)lk * per + e2];
      float* o3 = g.cpl == 5 ? out.A4 : out.A3;   // head factor
      float* o4 = g.cpl == 5 ? out.A5 : out.A4;   // head-dim factor
      (is3 ? o3 : o4)[e2] = W(g.s * cp.R1[r] * acc);
    }
    return;
  }
  __shared__ float red1[256], red2[256];
  // R <= 64: 256 / Rr groups of Rr lanes (Rr = R rounded up to 16 / 32 / 64) walk the 12L (layer, slot) pairs -- sixteen groups at
  // rank <= 16, nine pairs each (four groups of 64 lanes, 16 of them active, walked 36 pairs each: 27 us of dependent latencies)
  const int Rr = R <= 16 ? 16 : (R <= 32 ? 32 : 64), NG = 256 / Rr;
  const int r = threadIdx.x % Rr, qg = threadIdx.x / Rr;
  __shared__ float red3[3][256];
  float d1 = 0.f, d2 = 0.f, dk[3] = {0.f, 0.f, 0.f};
  if (r < R) {
#pragma unroll 3
    for (int q = qg; q < 12 * L; q += NG) {
      const float* zp = sc.zp + (size_t)q * GS_SPLIT * R + r;
      float z = 0.f;
#pragma unroll
      for (int sp = 0; sp < GS_SPLIT; ++sp) z += zp[sp * R];
      const int l = q / 12, slot = q - 12 * l;
      if (slot < 3) {
        if (g.cpl == 5 || g.cpl == 2) continue;   // order 5: A1 [depth, R] and A2 [3, R] mix the three projections, below; order 2: dense_delta.hip
        const int row = 3 * l + slot;
        d1 += cp.A1[row * R + r] * z;
        out.A1[row * R + r] = W(cp.R1[r] * z);
      } else {
        const int row = 9 * l + (slot - 3);   // 3 -> 9l ; 4..7 -> 9l+1..4 ; 8..11 -> 9l+5..8
        if (slot < 8) {
          d2 += cp.P1[row * R + r] * z;
          out.P1[row * R + r] = W(cp.R2[r] * z);
        } else {
          out.P1[row * R + r] = W(z);
        }
      }
    }
    if (g.cpl == 5) {
      // dA1[l] = R1 sum_k A2[k] Z[l,k] ;  dA2[k] = R1 sum_l A1[l] Z[l,k] ;  dR1 = sum_{l,k} A1[l] A2[k] Z[l,k]
      for (int l = qg; l < L; l += NG) {
        float a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* zp = sc.zp + (size_t)(l * 12 + k) * GS_SPLIT * R + r;
          float z = 0.f;
#pragma unroll
          for (int sp = 0; sp < GS_SPLIT; ++sp) z += zp[sp * R];
          a1 += cp.A2[k * R + r] * z;
          dk[k] += cp.A1[l * R + r] * z;
        }
        out.A1[l * R + r] = W(cp.R1[r] * a1);
        d1 += cp.A1[l * R + r] * a1;
      }
    }
    {
      float zv = 0.f;
      for (int i = qg; i < L * GS_SPLIT; i += NG) zv += sc.zvp[(size_t)i * R + r];
      d2 += g.s * zv;
    }
  }
  red1[threadIdx.x] = d1;
  red2[threadIdx.x] = d2;
#pragma unroll
  for (int k = 0; k < 3; ++k) red3[k][threadIdx.x] = dk[k];
  __syncthreads();
  if (qg == 0 && r < R) {   // the groups' sums in a fixed order
    float s1 = 0.f, s2 = 0.f, s3[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < NG; ++k) {
      s1 += red1[k * Rr + r];
      s2 += red2[k * Rr + r];
#pragma unroll
      for (int j = 0; j < 3; ++j) s3[j] += red3[j][k * Rr + r];
    }
    out.R1[r] = W(s1);
    out.R2[r] = W(s2);
    if (g.cpl == 5) {
#pragma unroll
      for (int k = 0; k < 3; ++k) out.A2[k * R + r] = W(cp.R1[r] * s3[k]);
    }
  }
}

PackDims dims_of(const cara_geom* g) {
  PackDims d;
  d.depth = g->depth; d.dim = g->dim; d.heads = g->heads; d.hd = g->dim / g->heads;
  d.rank = g->rank; d.Rp = g->Rp; d.s = g->scale;
  d.cpl = g->cp_length == 0 ? 4 : g->cp_length;
  d.loss_scale = nullptr;
  d.found_inf = nullptr;
  return d;
}
bool geom_ok(const cara_geom* g) {
  return g && g->depth > 0 && g->dim > 0 && g->dim % 8 == 0 && g->heads > 0 && g->dim % g->heads == 0 && g->rank > 0 &&
         g->rank <= g->Rp && (g->Rp == 32 || g->Rp == 64) &&
         (g->cp_length == 0 || g->cp_length == 2 || g->cp_length == 3 || g->cp_length == 4 || g->cp_length == 5);
}
bool cp_ok(const cara_geom* g, const cara_cp* c) {
  if (!(c && c->A1 && c->A2 && c->P1 && c->P2 && c->P3 && c->R1 && c->R2 && c->bias1 && c->bias2 && c->bias3)) return false;
  if (g->cp_length == 2) return true;              // order 2: A1 [3 depth, R] and A2 [dim * dim, R] only
  if (!c->A3) return false;
  if (g->cp_length != 3 && !c->A4) return false;   // order 3 has no fourth factor
  return g->cp_length != 5 || c->A5 != nullptr;
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
  if (!geom_ok(g) || !cp_ok(g, cp) || !base_bias_proj || !base_bias_fc1 || !base_bias_fc2 || !pack) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackDims d = dims_of(g);
  const PackOffsets po = make_offsets(g->dim, g->Rp);
  const int maxn = 2 * 4 * g->dim * g->Rp / 8;   // threads of eight elements, both copies of the tallest matrix
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
  return (3 * L * (H + hd) * R + L * GS_SPLIT * R + 12 * L * GS_SPLIT * R) * sizeof(float);
}

extern "C" int cara_factor_grad_reduce(const cara_geom* g, const cara_cp* cp, const cara_layer_grads* lg,
                                       const cara_cp* grads, void* scratch, void* stream) {
  return cara_factor_grad_reduce_ex(g, cp, lg, grads, scratch, nullptr, nullptr, stream);
}
extern "C" int cara_factor_grad_reduce_ex(const cara_geom* g, const cara_cp* cp, const cara_layer_grads* lg,
                                          const cara_cp* grads, void* scratch, const float* loss_scale, float* found_inf,
                                          void* stream) {
  if (!geom_ok(g) || !cp_ok(g, cp) || !cp_ok(g, grads) || !lg || !scratch) return CARA_E_ARG;
  if (!lg->dU_qkv || !lg->dVs_qkv || !lg->dU_proj || !lg->dVs_proj || !lg->dU_fc1 || !lg->dVs_fc1 || !lg->dU_fc2 ||
      !lg->dVs_fc2 || !lg->dc_proj || !lg->dc_fc1 || !lg->dc_fc2)
    return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PackDims d = dims_of(g);
  d.loss_scale = loss_scale;
  d.found_inf = found_inf;
  const int n1 = g->dim * g->rank > 4 * g->dim ? g->dim * g->rank : 4 * g->dim;
  const int nb_row = (n1 + ROW_E - 1) / ROW_E, nb_col = g->depth * 13 * GS_SPLIT, nb_a34 = 3 * g->depth * 5;
  float* sc = static_cast<float*>(scratch);
  static const int split = [] { const char* e = getenv("CARA_GRAD_STAGE1_SPLIT"); return e ? atoi(e) : 0; }();
  if (split) {
    hipLaunchKernelGGL(grad_stage1_kernel, dim3(nb_row), dim3(256), 0, st, d, *cp, *lg, *grads, sc, nb_row, 0);
    hipLaunchKernelGGL(grad_stage1_kernel, dim3(nb_col), dim3(256), 0, st, d, *cp, *lg, *grads, sc, nb_row, nb_row);
    hipLaunchKernelGGL(grad_stage1_kernel, dim3(nb_a34), dim3(256), 0, st, d, *cp, *lg, *grads, sc, nb_row, nb_row + nb_col);
  } else {
    hipLaunchKernelGGL(grad_stage1_kernel, dim3(nb_row + nb_col + nb_a34), dim3(256), 0, st, d, *cp, *lg, *grads, sc, nb_row, 0);
  }
  CARA_CHECK_LAUNCH();
  const int nb_a = ((g->heads + g->dim / g->heads) * g->rank + 255) / 256;
  hipLaunchKernelGGL(grad_stage2_kernel, dim3(nb_a + 1), dim3(256), 0, st, d, *cp, sc, *grads, nb_a);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
