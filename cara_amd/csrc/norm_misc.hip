// LayerNorm forward/backward and the small pieces of the ViT around the blocks (gfx950 only).
// timm 0.4.12 Block: x = x + drop_path(attn(norm1(x))); x = x + drop_path(mlp(norm2(x)))
// (restated in oracle/cara_oracle.py).  The residual stream stays fp32; normalised activations
// leave as bf16 GEMM operands.  All kernels are HBM-bound streaming passes: one wave per row,
// 16-byte accesses, wave shuffles for the two row reductions.
#include "common.h"

namespace {

// Skinny adapter contraction fused into the kernel that PRODUCES its input (T = X U in
// the forward, G' = dY Vs in the backward; K = C columns, Rp = 32 output columns of which the first `rank`
// are non-zero).  A block of 8 waves normalises 16 rows (2 each), leaves their bf16 images in LDS
// ([16][C + 8]: the 16-byte pad spreads the rows of an MFMA A fragment over the banks), then contracts them
// with Ut [32][C] on the matrix cores: wave w takes the K steps w, w+8, ... (its B fragments come straight
// from global / L2: 1.5 KB per row, the cost of the staging the separate cara_skinny_xu pass pays too), the
// eight partial 16 x 32 tiles are summed through LDS in fixed order.  Two earlier forms were measured and
// dropped: per-lane v_dot2 sums with the factor slice read per row (more L1 traffic than the pass it
// replaces) or held in LDS (as slow as LayerNorm + cara_skinny_xu: ~250 VALU instructions per row).
constexpr int XU_WAVES = 8;         // waves per block of the fused kernels
constexpr int XU_ROWS = 16 / XU_WAVES;   // rows per wave: 16 per block (one MFMA row tile)
// NT = Rp / 16 column tiles of the contraction: 2 (rank <= 32) or 4 (rank <= 64)
template <int V4, int NT = 2>
struct XuLds {
  static constexpr int LDY = V4 * 256 + 8;   // bf16 elements per staged row
  bf16 y[16 * LDY];
  float part[XU_WAVES][NT][64 * 4];
};
// the block's 16 staged bf16 rows -> K-panel-major global image [C/32][yp][32]: a wave writes one panel's 16 rows
// (16 x 64 B = one contiguous KiB of full cache lines) per instruction, instead of each row's 8-byte pieces
// scattered over C/32 panels.  Call after the rows are complete (a __syncthreads() away from the staging).
template <int V4, int NT>
__device__ __forceinline__ void panel_store(const XuLds<V4, NT>& L, bf16* __restrict__ y, const int yp, const int row_base, const int M) {
  constexpr int C = V4 * 256, LDY = XuLds<V4, NT>::LDY;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 2, chunk = lane & 3;
  if (row_base + row >= M) return;
#pragma unroll
  for (int p = wave; p < C / 32; p += XU_WAVES)
    *reinterpret_cast<bf16x8*>(y + ((size_t)p * yp + row_base + row) * 32 + chunk * 8) =
        *reinterpret_cast<const bf16x8*>(L.y + row * LDY + p * 32 + chunk * 8);
}
// a wave's B fragments of the contraction (its K steps wave, wave + 8, ...: C/256 of them, two column tiles each):
// requested at the top of the kernel so that their L2 latency passes under the row loads and reductions
template <int V4, int NT = 2>
struct UtFrags {
  bf16x8 u[V4][NT];
};
template <int V4, int NT>
__device__ __forceinline__ UtFrags<V4, NT> load_ut_frags(const bf16* __restrict__ Ut) {
  constexpr int C = V4 * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  UtFrags<V4, NT> f;
#pragma unroll
  for (int k = 0; k < V4; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      f.u[k][nt] = *reinterpret_cast<const bf16x8*>(Ut + (size_t)(nt * 16 + fr) * C + (wave + k * XU_WAVES) * 32 + fq * 8);
  return f;
}
template <int V4, int NT>
__device__ __forceinline__ void block_contract(XuLds<V4, NT>& L, const UtFrags<V4, NT>& uf, bf16* __restrict__ T,
                                               bf16* __restrict__ Tt, const int ldt, const int row_base, const int M, const int Rp) {
  constexpr int LDY = XuLds<V4, NT>::LDY;
  static_assert(XU_WAVES == 8, "K steps per wave = C / 32 / 8 = V4");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  __syncthreads();   // the 16 staged rows are complete
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < V4; ++k) {
    const int s = wave + k * XU_WAVES;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(L.y + fr * LDY + s * 32 + fq * 8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, uf.u[k][nt], acc[nt], 0, 0, 0);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(&L.part[wave][nt][lane * 4]) = acc[nt];
  __syncthreads();
  if (wave < NT) {   // wave nt sums column tile nt: lane (fr, fq) holds rows fq*4 .. +3 of column nt*16 + fr
    const int nt = wave;
    f32x4 t = *reinterpret_cast<const f32x4*>(&L.part[0][nt][lane * 4]);
#pragma unroll
    for (int w = 1; w < XU_WAVES; ++w) t += *reinterpret_cast<const f32x4*>(&L.part[w][nt][lane * 4]);
    const int col = nt * 16 + fr, m0 = row_base + fq * 4;
    const bf16x4 o = {(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m0 + r < M) T[(size_t)(m0 + r) * Rp + col] = o[r];
    if (Tt) {
      if (m0 + 4 <= M) {
        *reinterpret_cast<bf16x4*>(Tt + (size_t)col * ldt + m0) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + r < M) Tt[(size_t)col * ldt + m0 + r] = o[r];
      }
    }
  }
  else if (wave < Rp / 16) {
    // NT * 16 < Rp (rank <= 16 at Rp = 32: only the first column tile is computed, the rows of Ut beyond the rank are
    // zero): the tiles that were skipped are written as the zeros they are
    const int col = wave * 16 + fr, m0 = row_base + fq * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (m0 + r >= M) continue;
      T[(size_t)(m0 + r) * Rp + col] = (bf16)0.f;
      if (Tt) Tt[(size_t)col * ldt + m0 + r] = (bf16)0.f;
    }
  }
}

template <int V4, bool XU, int NT = 2>  // V4 = C / 256 : float4 per lane; XU: fused contraction, XU_ROWS rows per wave
__global__ __launch_bounds__(XU ? XU_WAVES * 64 : 256) void ln_fwd_kernel(const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     bf16* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, float eps,
                                                     const bf16* __restrict__ Ut, int rank, int Rp, bf16* __restrict__ T,
                                                     bf16* __restrict__ Tt, int ldt, const int yp) {
  constexpr int C = V4 * 256;
  constexpr int RPW = XU ? XU_ROWS : 1;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (XU ? XU_WAVES : 4) + (threadIdx.x >> 6)) * RPW;
  __shared__ __attribute__((aligned(16))) XuLds<XU ? V4 : 0, XU ? NT : 1> L;
  // (NT = 4: twice the fragments; they are requested behind the row loop, as in the backward kernel, so that the kernel
  // keeps its occupancy)
  UtFrags<XU ? V4 : 1, XU ? NT : 1> uf;
  if constexpr (XU && NT <= 2) uf = load_ut_frags<V4, NT>(Ut);
  // all of a wave's rows are requested before anything is reduced: a wave keeps RPW x V4 16-byte loads in flight
  // instead of V4 (the kernel is latency-bound: one row at a time reached 3.4 TB/s)
  float4 v[RPW][V4];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr < M ? row0 + rr : M - 1;
    const float* xr = x + (size_t)row * ldx;
#pragma unroll
    for (int i = 0; i < V4; ++i) v[rr][i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
  }
  float4 gm[V4], bt[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    gm[i] = *reinterpret_cast<const float4*>(gamma + i * 256 + lane * 4);
    bt[i] = *reinterpret_cast<const float4*>(beta + i * 256 + lane * 4);
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
  const int row = row0 + rr;
  if (row >= M) break;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) s += (v[rr][i].x + v[rr][i].y) + (v[rr][i].z + v[rr][i].w);
  const float mu = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const float a = v[rr][i].x - mu, b = v[rr][i].y - mu, c = v[rr][i].z - mu, d = v[rr][i].w - mu;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rs = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
  bf16* yr = y + (size_t)row * C;
  bf16x4 yb[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int c0 = i * 256 + lane * 4;
    const float4 g = gm[i];
    const float4 b = bt[i];
    bf16x4 o = {(bf16)((v[rr][i].x - mu) * rs * g.x + b.x), (bf16)((v[rr][i].y - mu) * rs * g.y + b.y),
                (bf16)((v[rr][i].z - mu) * rs * g.z + b.z), (bf16)((v[rr][i].w - mu) * rs * g.w + b.w)};
    // yp > 0: y as K-panel-major [C/32][yp][32] (the A operand layout of the GEMM that reads it)
    if (!(XU && yp))   // (fused kernels write the panel image from the staged rows, panel_store)
      *reinterpret_cast<bf16x4*>(yp ? y + ((size_t)(c0 >> 5) * yp + row) * 32 + (c0 & 31) : yr + c0) = o;
    yb[i] = o;
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  if constexpr (XU) {
#pragma unroll
    for (int i = 0; i < V4; ++i)
      *reinterpret_cast<bf16x4*>(L.y + (row - blockIdx.x * 16) * XuLds<V4, NT>::LDY + i * 256 + lane * 4) = yb[i];
  }
  }
  if constexpr (XU) {
    if (yp) {
      __syncthreads();
      panel_store<V4, NT>(L, y, yp, blockIdx.x * 16, M);
    }
    if constexpr (NT > 2) uf = load_ut_frags<V4, NT>(Ut);
    block_contract<V4, NT>(L, uf, T, Tt, ldt, blockIdx.x * 16, M, Rp);   // T = LN(x) U of the next linear
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mu) * rstd
template <int V4, bool XU, int NT = 2>
__global__ __launch_bounds__(XU ? XU_WAVES * 64 : 256) void ln_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                     long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ dx_in,
                                                     float* __restrict__ dx_out, bf16* __restrict__ dyb,
                                                     const float* __restrict__ rowscale,
                                                     int rows_per_sample, int M,
                                                     const bf16* __restrict__ Vst, int rank, int Rp, bf16* __restrict__ G,
                                                     bf16* __restrict__ Gt, int ldt, const int yp, const int dx_in_every) {
  // dx_in_every > 0: the running gradient dx_in is nonzero on rows that are multiples of it only (the last block of a ViT whose head
  // reads the cls token: gradient has reached nothing but the cls rows yet) -- the other rows are neither read nor expected zeroed
  constexpr int C = V4 * 256;
  constexpr int RPW = XU ? XU_ROWS : 1;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (XU ? XU_WAVES : 4) + (threadIdx.x >> 6)) * RPW;
  __shared__ __attribute__((aligned(16))) XuLds<XU ? V4 : 0, XU ? NT : 1> L;
  // Every load of the wave's rows -- x, dy AND the running gradient dx_in, which the second half of a row's work reads -- is
  // requested before anything is reduced: one memory latency per wave instead of two per row (the kernel is one round of
  // waves, all in the same phase: nothing else hides a dependent load)
  float4 xv[RPW][V4], pv[RPW][V4];
  bf16x4 dv[RPW][V4];
  float mu_[RPW], rs_[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr < M ? row0 + rr : M - 1;
    const float* xr = x + (size_t)row * ldx;
    const bf16* dr = dy + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      xv[rr][i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
      dv[rr][i] = *reinterpret_cast<const bf16x4*>(dr + i * 256 + lane * 4);
    }
    mu_[rr] = mean[row];
    rs_[rr] = rstd[row];
  }
  bool has_in[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr < M ? row0 + rr : M - 1;
    has_in[rr] = dx_in && (dx_in_every <= 0 || row % dx_in_every == 0);   // (wave-uniform: a wave owns whole rows)
    if (has_in[rr]) {
#pragma unroll
      for (int i = 0; i < V4; ++i) pv[rr][i] = *reinterpret_cast<const float4*>(dx_in + (size_t)row * ldx + i * 256 + lane * 4);
    }
  }
  float4 gmv[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) gmv[i] = *reinterpret_cast<const float4*>(gamma + i * 256 + lane * 4);
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
  const int row = row0 + rr;
  if (row >= M) break;
  const float mu = mu_[rr], rs = rs_[rr];
  float4 g[V4], xh[V4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const float4 xq = xv[rr][i];
    const float4 gm = gmv[i];
    const bf16x4 d = dv[rr][i];
    xh[i] = make_float4((xq.x - mu) * rs, (xq.y - mu) * rs, (xq.z - mu) * rs, (xq.w - mu) * rs);
    g[i] = make_float4((float)d[0] * gm.x, (float)d[1] * gm.y, (float)d[2] * gm.z, (float)d[3] * gm.w);
    s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
    s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
  }
  const float c1 = wave_sum(s1) * (1.0f / C), c2 = wave_sum(s2) * (1.0f / C);
  const float sc = rowscale ? rowscale[row / rows_per_sample] : 1.f;
  float* dor = dx_out + (size_t)row * ldx;
  bf16* db = dyb ? dyb + (size_t)row * ldx : nullptr;
  bf16x4 yb[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int c0 = i * 256 + lane * 4;
    float4 o = make_float4(rs * (g[i].x - c1 - xh[i].x * c2), rs * (g[i].y - c1 - xh[i].y * c2),
                           rs * (g[i].z - c1 - xh[i].z * c2), rs * (g[i].w - c1 - xh[i].w * c2));
    if (has_in[rr]) {
      const float4 pq = pv[rr][i];
      o.x += pq.x; o.y += pq.y; o.z += pq.z; o.w += pq.w;
    }
    *reinterpret_cast<float4*>(dor + c0) = o;
    bf16x4 b = {(bf16)(o.x * sc), (bf16)(o.y * sc), (bf16)(o.z * sc), (bf16)(o.w * sc)};
    if (db && !(XU && yp)) *reinterpret_cast<bf16x4*>(yp ? dyb + ((size_t)(c0 >> 5) * yp + row) * 32 + (c0 & 31) : db + c0) = b;
    yb[i] = b;
  }
  if constexpr (XU) {
#pragma unroll
    for (int i = 0; i < V4; ++i)
      *reinterpret_cast<bf16x4*>(L.y + (row - blockIdx.x * 16) * XuLds<V4, NT>::LDY + i * 256 + lane * 4) = yb[i];
  }
  }
  if constexpr (XU) {
    if (yp) {
      __syncthreads();
      panel_store<V4, NT>(L, dyb, yp, blockIdx.x * 16, M);
    }
    // (the B fragments are requested here, not at the top: holding them through the row loop costs the backward kernel a
    // workgroup of occupancy -- 138 VGPRs, 36 us instead of 31)
    block_contract<V4, NT>(L, load_ut_frags<V4, NT>(Vst), G, Gt, ldt, blockIdx.x * 16, M, Rp);   // G' = dY Vs of the linear below
  }
}

// images fp32 [B,C,Hi,Wi] -> patch rows bf16 [B*gh*gw, C*p*p], column = (c*p + py)*p + px
// (the flattening of Conv2d weight [D, C, p, p]); p == 16: one 16-float image row segment per
// 4 lanes.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out,
                                                     int B, int C, int Hi, int Wi, int p, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of a patch row
  if (idx >= total4) return;
  const int gw = Wi / p, gh = Hi / p;
  const int kcols = C * p * p;
  const long e = idx * 4;
  const long row = e / kcols;
  const int col = (int)(e - row * kcols);
  const int c = col / (p * p), py = (col / p) % p, px = col % p;
  const int b = (int)(row / (gh * gw)), pr = (int)(row % (gh * gw));
  const int gy = pr / gw, gx = pr % gw;
  const float4 v = *reinterpret_cast<const float4*>(img + (((size_t)b * C + c) * Hi + gy * p + py) * Wi + gx * p + px);
  bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
  *reinterpret_cast<bf16x4*>(out + e) = o;
}

__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, float* __restrict__ x,
                                                       int B, int P, int D, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const long e = idx * 4;
  const int d = (int)(e % D);
  const long tok = e / D;
  const int t = (int)(tok % (P + 1)), b = (int)(tok / (P + 1));
  const float4 pv = *reinterpret_cast<const float4*>(pos + (size_t)t * D + d);
  float4 s = (t == 0) ? *reinterpret_cast<const float4*>(cls + d)
                      : *reinterpret_cast<const float4*>(emb + ((size_t)b * P + (t - 1)) * D + d);
  s.x += pv.x; s.y += pv.y; s.z += pv.z; s.w += pv.w;
  *reinterpret_cast<float4*>(x + e) = s;
}

// one wave per sample; loss = mean_b (lse_b - logit_b[y_b]); dlogits = (softmax - onehot) / B
__global__ __launch_bounds__(64) void xent_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  float* __restrict__ loss_per, float* __restrict__ dlogits,
                                                  int B, int C, float dscale, const float* __restrict__ loss_scale) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* lr = logits + (size_t)b * C;
  float m = -3.0e38f;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, lr[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(lr[c] - m);
  s = wave_sum(s);
  const float lse = m + __logf(s);
  const int y = (int)labels[b];
  const float invB = 1.0f / B;
  // (the gradient's scale: 1/B of the mean, times 1/world of a data-parallel job, times the loss scale of the IEEE-half build)
  const float gsc = invB * dscale * (loss_scale ? *loss_scale : 1.f);
  if (dlogits)
    for (int c = lane; c < C; c += 64)
      dlogits[(size_t)b * C + c] = (__expf(lr[c] - lse) - (c == y ? 1.f : 0.f)) * gsc;
  if (lane == 0) loss_per[b] = (lse - lr[y]) * invB;
}
__global__ __launch_bounds__(64) void xent_sum_kernel(const float* __restrict__ loss_per, float* __restrict__ loss, int B,
                                                      float* __restrict__ found_inf) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += loss_per[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    *loss = s;
    if (found_inf) *found_inf = 0.f;   // raised by the backward's final gradient writes (cara_vit_shape::found_inf)
  }
}

__global__ __launch_bounds__(256) void cvt_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
    *reinterpret_cast<bf16x4*>(dst + i) = o;
  } else {
    for (size_t k = i; k < n; ++k) dst[k] = (bf16)src[k];
  }
}

// dst[c][r] = src[r][c], 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                        int rows, int cols) {
  __shared__ bf16 tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = src[(size_t)(r0 + k) * cols + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) dst[(size_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// dst[c * ldd + r] = src[r * lds + c] for a [rows, cols] matrix, 64 x 64 tiles, 16-byte global accesses both ways
// (activation-sized transposes of the exact weight-dropout mode: dY^T and X^T feed the dense dW GEMM)
__global__ __launch_bounds__(256) void transpose64_kernel(const bf16* __restrict__ src, long lds_, bf16* __restrict__ dst, long ldd,
                                                          int rows, int cols) {
  __shared__ bf16 tile[64][72];
  const int t = threadIdx.x, c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (t >> 3) + 32 * j, c = (t & 7) * 8;
    bf16x8 v = {};
    if (r0 + r < rows) {
      if (c0 + c + 8 <= cols) {
        v = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + r) * lds_ + c0 + c);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (c0 + c + k < cols) v[k] = src[(size_t)(r0 + r) * lds_ + c0 + c + k];
      }
    }
    *reinterpret_cast<bf16x8*>(&tile[r][c]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (t >> 3) + 32 * j, r = (t & 7) * 8;   // output row c0 + c, 8 consecutive source rows
    if (c0 + c >= cols) continue;
    bf16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tile[r + k][c];
    bf16* d = dst + (size_t)(c0 + c) * ldd + r0 + r;
    if (r0 + r + 8 <= rows) {
      *reinterpret_cast<bf16x8*>(d) = v;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r0 + r + k < rows) d[k] = v[k];
    }
  }
}

}  // namespace

namespace {
struct XuArgs {   // optional fused skinny product (all NULL / 0 = plain LayerNorm)
  const bf16* Ut;
  int rank, Rp;
  bf16 *T, *Tt;
  int ldt;
};
bool xu_ok(const XuArgs& a, int M, int C) {
  if (!a.Ut) return true;
  // Rp / 16 column tiles of 16 (rows of Ut beyond the rank are zero, so the pad columns come out zero)
  return a.T && (a.Rp == 32 || a.Rp == 64) && a.rank > 0 && a.rank <= a.Rp && (!a.Tt || (a.ldt >= M && !(a.ldt & 7))) &&
         (C == 768 || C == 256 || C == 1024);
}
// consumers (cara_tskinny_*) read Tt in whole 32-row steps: keep columns [M, roundup32(M)) zero (as cara_skinny_xu does)
int xu_pad(const XuArgs& a, int M, hipStream_t st) {
  const int m32 = (M + 31) / 32 * 32;
  if (a.Ut && a.Tt && m32 > M && m32 <= a.ldt &&
      hipMemset2DAsync(a.Tt + M, (size_t)a.ldt * 2, 0, (size_t)(m32 - M) * 2, a.Rp, st) != hipSuccess)
    return CARA_E_LAUNCH;
  return CARA_OK;
}

int ln_fwd_launch(const float* x, long ldx, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                  int C, float eps, const XuArgs& a, void* stream, int yp = 0) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || ldx < C || (ldx & 3) || !xu_ok(a, M, C)) return CARA_E_ARG;
  if (yp && yp < M) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (xu_pad(a, M, st) != CARA_OK) return CARA_E_LAUNCH;
  const int rows_per_block = a.Ut ? 16 : 4;
  const dim3 grid((M + rows_per_block - 1) / rows_per_block), block(a.Ut ? XU_WAVES * 64 : 256);
#define LNF(V, X) hipLaunchKernelGGL((ln_fwd_kernel<V, X>), grid, block, 0, st, x, ldx, gamma, beta, (bf16*)y, mean, rstd, M, eps, \
                                     a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp)
#define LNF4(V) hipLaunchKernelGGL((ln_fwd_kernel<V, true, 4>), grid, block, 0, st, x, ldx, gamma, beta, (bf16*)y, mean, rstd, M, eps, \
                                   a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp)
#define LNF1(V) hipLaunchKernelGGL((ln_fwd_kernel<V, true, 1>), grid, block, 0, st, x, ldx, gamma, beta, (bf16*)y, mean, rstd, M, eps, \
                                   a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp)
  if (a.Ut && a.Rp == 64) {
    if (C == 768) LNF4(3);
    else if (C == 1024) LNF4(4);
    else LNF4(1);
  } else if (a.Ut && a.rank <= 16) {   // half of the padded rank is structurally zero: one column tile of the contraction
    if (C == 768) LNF1(3);
    else if (C == 1024) LNF1(4);
    else LNF1(1);
  } else if (a.Ut) {
    if (C == 768) LNF(3, true);
    else if (C == 1024) LNF(4, true);
    else LNF(1, true);
  } else if (C == 768) LNF(3, false);
  else if (C == 1024) LNF(4, false);
  else if (C == 256) LNF(1, false);
  else return CARA_E_ARG;
#undef LNF
#undef LNF4
#undef LNF1
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

int ln_bwd_launch(const void* dy, const float* x, long ldx, const float* gamma, const float* mean, const float* rstd,
                  const float* dx_in, float* dx_out, void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                  const XuArgs& a, void* stream, int yp = 0, int dx_in_every = 0) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx_out || M <= 0 || ldx < C || (ldx & 3) || !xu_ok(a, M, C)) return CARA_E_ARG;
  if (yp && (yp < M || !dyb)) return CARA_E_ARG;
  if (rowscale && rows_per_sample <= 0) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (xu_pad(a, M, st) != CARA_OK) return CARA_E_LAUNCH;
  const int rows_per_block = a.Ut ? 16 : 4;
  const dim3 grid((M + rows_per_block - 1) / rows_per_block), block(a.Ut ? XU_WAVES * 64 : 256);
#define LNB(V, X) hipLaunchKernelGGL((ln_bwd_kernel<V, X>), grid, block, 0, st, (const bf16*)dy, x, ldx, gamma, mean, rstd, \
                                     dx_in, dx_out, (bf16*)dyb, rowscale, rows_per_sample, M, a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp, dx_in_every)
#define LNB4(V) hipLaunchKernelGGL((ln_bwd_kernel<V, true, 4>), grid, block, 0, st, (const bf16*)dy, x, ldx, gamma, mean, rstd, \
                                   dx_in, dx_out, (bf16*)dyb, rowscale, rows_per_sample, M, a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp, dx_in_every)
#define LNB1(V) hipLaunchKernelGGL((ln_bwd_kernel<V, true, 1>), grid, block, 0, st, (const bf16*)dy, x, ldx, gamma, mean, rstd, \
                                   dx_in, dx_out, (bf16*)dyb, rowscale, rows_per_sample, M, a.Ut, a.rank, a.Rp, a.T, a.Tt, a.ldt, yp, dx_in_every)
  if (a.Ut && a.Rp == 64) {
    if (C == 768) LNB4(3);
    else if (C == 1024) LNB4(4);
    else LNB4(1);
  } else if (a.Ut && a.rank <= 16) {
    if (C == 768) LNB1(3);
    else if (C == 1024) LNB1(4);
    else LNB1(1);
  } else if (a.Ut) {
    if (C == 768) LNB(3, true);
    else if (C == 1024) LNB(4, true);
    else LNB(1, true);
  } else if (C == 768) LNB(3, false);
  else if (C == 1024) LNB(4, false);
  else if (C == 256) LNB(1, false);
  else return CARA_E_ARG;
#undef LNB
#undef LNB4
#undef LNB1
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
}  // namespace

extern "C" int cara_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int M, int C, float eps, void* stream) {
  return ln_fwd_launch(x, ldx, gamma, beta, y, mean, rstd, M, C, eps, XuArgs{nullptr, 0, 0, nullptr, nullptr, 0}, stream);
}
extern "C" int cara_layernorm_fwd_xu(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                                     float* mean, float* rstd, int M, int C, float eps, const void* Ut, int rank, int Rp,
                                     void* T, void* Tt, int ldt, void* stream) {
  if (!Ut) return CARA_E_ARG;
  return ln_fwd_launch(x, ldx, gamma, beta, y, mean, rstd, M, C, eps,
                       XuArgs{static_cast<const bf16*>(Ut), rank, Rp, static_cast<bf16*>(T), static_cast<bf16*>(Tt), ldt}, stream);
}
extern "C" int cara_layernorm_bwd(const void* dy, const float* x, long ldx, const float* gamma, const float* mean,
                                  const float* rstd, const float* dx_in, float* dx_out, void* dyb,
                                  const float* rowscale, int rows_per_sample, int M, int C, void* stream) {
  return ln_bwd_launch(dy, x, ldx, gamma, mean, rstd, dx_in, dx_out, dyb, rowscale, rows_per_sample, M, C,
                       XuArgs{nullptr, 0, 0, nullptr, nullptr, 0}, stream);
}
extern "C" int cara_layernorm_bwd_xu(const void* dy, const float* x, long ldx, const float* gamma, const float* mean,
                                     const float* rstd, const float* dx_in, float* dx_out, void* dyb,
                                     const float* rowscale, int rows_per_sample, int M, int C, const void* Vst, int rank,
                                     int Rp, void* G, void* Gt, int ldt, void* stream) {
  if (!Vst) return CARA_E_ARG;
  return ln_bwd_launch(dy, x, ldx, gamma, mean, rstd, dx_in, dx_out, dyb, rowscale, rows_per_sample, M, C,
                       XuArgs{static_cast<const bf16*>(Vst), rank, Rp, static_cast<bf16*>(G), static_cast<bf16*>(Gt), ldt}, stream);
}
// The general forms: Ut may be NULL (no fused contraction), and the bf16 output (y, resp. dyb) can be written
// K-panel-major -- [C/32][panels][32], panels >= M rows per panel -- the layout cara_gemm_args::a_panels reads.
extern "C" int cara_layernorm_fwd_ex(const float* x, long ldx, const float* gamma, const float* beta, void* y,
                                     float* mean, float* rstd, int M, int C, float eps, const void* Ut, int rank, int Rp,
                                     void* T, void* Tt, int ldt, int y_panels, void* stream) {
  if (y_panels < 0) return CARA_E_ARG;
  const XuArgs a = Ut ? XuArgs{static_cast<const bf16*>(Ut), rank, Rp, static_cast<bf16*>(T), static_cast<bf16*>(Tt), ldt}
                      : XuArgs{nullptr, 0, 0, nullptr, nullptr, 0};
  return ln_fwd_launch(x, ldx, gamma, beta, y, mean, rstd, M, C, eps, a, stream, y_panels);
}
extern "C" int cara_layernorm_bwd_ex(const void* dy, const float* x, long ldx, const float* gamma, const float* mean,
                                     const float* rstd, const float* dx_in, float* dx_out, void* dyb, const float* rowscale,
                                     int rows_per_sample, int M, int C, const void* Vst, int rank, int Rp, void* G, void* Gt,
                                     int ldt, int dyb_panels, void* stream) {
  if (dyb_panels < 0) return CARA_E_ARG;
  const XuArgs a = Vst ? XuArgs{static_cast<const bf16*>(Vst), rank, Rp, static_cast<bf16*>(G), static_cast<bf16*>(Gt), ldt}
                       : XuArgs{nullptr, 0, 0, nullptr, nullptr, 0};
  return ln_bwd_launch(dy, x, ldx, gamma, mean, rstd, dx_in, dx_out, dyb, rowscale, rows_per_sample, M, C, a, stream, dyb_panels);
}

// cara_layernorm_bwd_ex with dx_in read on every dx_in_every-th row only (not in include/cara_hip.h: vit.hip's last block)
int cara_layernorm_bwd_rows_in(const void* dy, const float* x, long ldx, const float* gamma, const float* mean, const float* rstd,
                               const float* dx_in, float* dx_out, void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                               const void* Vst, int rank, int Rp, void* G, void* Gt, int ldt, int dyb_panels, int dx_in_every,
                               void* stream) {
  if (dyb_panels < 0 || dx_in_every < 0) return CARA_E_ARG;
  const XuArgs a = Vst ? XuArgs{static_cast<const bf16*>(Vst), rank, Rp, static_cast<bf16*>(G), static_cast<bf16*>(Gt), ldt}
                       : XuArgs{nullptr, 0, 0, nullptr, nullptr, 0};
  return ln_bwd_launch(dy, x, ldx, gamma, mean, rstd, dx_in, dx_out, dyb, rowscale, rows_per_sample, M, C, a, stream, dyb_panels, dx_in_every);
}

extern "C" int cara_im2col_patches(const float* img, void* patches, int B, int C, int Hi, int Wi, int p, void* stream) {
  if (!img || !patches || B <= 0 || C <= 0 || p <= 0 || (p & 3) || Hi % p || Wi % p || (Wi & 3)) return CARA_E_ARG;
  const long total4 = (long)B * C * Hi * Wi / 4;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     img, (bf16*)patches, B, C, Hi, Wi, p, total4);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_assemble_tokens(const float* emb, const float* cls, const float* pos, float* x, int B, int P,
                                    int D, void* stream) {
  if (!emb || !cls || !pos || !x || B <= 0 || P <= 0 || D <= 0 || (D & 3)) return CARA_E_ARG;
  const long total4 = (long)B * (P + 1) * D / 4;
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     emb, cls, pos, x, B, P, D, total4);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_cross_entropy_ex(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B,
                                     int C, float dscale, const float* loss_scale, float* found_inf, void* stream) {
  // loss doubles as scratch: needs room for 1 + B floats (loss[0] = mean loss, loss[1..B] per-sample terms)
  if (!logits || !labels || !loss || B <= 0 || C <= 0 || !(dscale > 0.f)) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(xent_kernel, dim3(B), dim3(64), 0, st, logits, labels, loss + 1, dlogits, B, C, dscale, loss_scale);
  CARA_CHECK_LAUNCH();
  hipLaunchKernelGGL(xent_sum_kernel, dim3(1), dim3(64), 0, st, loss + 1, loss, B, found_inf);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
extern "C" int cara_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, int B,
                                  int C, void* stream) {
  return cara_cross_entropy_ex(logits, labels, loss, dlogits, B, C, 1.f, nullptr, nullptr, stream);
}

extern "C" int cara_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
  if (!src || !dst || n == 0) return CARA_E_ARG;
  const size_t nthreads = (n + 3) / 4;
  hipLaunchKernelGGL(cvt_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, (bf16*)dst, n);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_transpose_bf16_ld(const void* src, long lds, void* dst, long ldd, int rows, int cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < rows || (lds & 7) || (ldd & 7)) return CARA_E_ARG;
  hipLaunchKernelGGL(transpose64_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const bf16*)src, lds, (bf16*)dst, ldd, rows, cols);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return CARA_E_ARG;
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const bf16*)src, (bf16*)dst, rows, cols);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
