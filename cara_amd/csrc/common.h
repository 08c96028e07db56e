// Device-side helpers shared by the gfx950 kernels of libcara_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cara_hip.h"

// The 16-bit operand type of the build.  Default: bf16 (8 significand bits), what BASELINE.json's metric is quoted on.
// -DCARA_F16_OPERANDS (build.sh f16 -> libcara_hip_f16.so, `precision = "fp16"` of the Python side): IEEE half, 11 significand
// bits at the SAME MFMA rate (v_mfma_f32_16x16x32_f16 / 32x32x16_f16) -- the same kernels, rounding points and layouts with
// ~8x less operand rounding error: the forward then meets north_star's 1e-3 on the logits (the oracle's
// fp16 rounding model: 9.2e-4 at depth 12), the backward runs with a static loss scale (half's range, not its precision, is
// what gradients need).  Every "bf16" in the sources and in include/cara_hip.h reads "the build's 16-bit operand type".
#ifdef CARA_F16_OPERANDS
typedef _Float16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define CARA_OPERAND_TYPE "fp16"
#else
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define CARA_OPERAND_TYPE "bf16"
#endif
// IEEE half whatever the build's operand type: the saved gelu'(u) of CARA_EPI_GELU_DG / CARA_EPI_MULH
typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(2))) float f32x2;
// Two fp32 values to ONE packed pair of the build's 16-bit type (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), returned as the dword.
// The empty asm keeps the pair packed: where the two halves are then used separately (2-byte LDS stores of an epilogue) hipcc
// otherwise converts IEEE halves one by one (v_cvt_f16_f32 x 2) -- the fp16 build's epilogues were 2-3 us per launch slower than
// the bf16 build's for that alone (profiles/r05_a_fp16_vs_bf16_sites.txt).
__device__ __forceinline__ unsigned cvt_pk_dword(float a, float b) {
  const f32x2 v = {a, b};
  unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  asm volatile("" : "+v"(u));
  return u;
}

// a[0] b[0] + a[1] b[1] + c in fp32, one instruction (v_dot2c_f32_bf16 / v_dot2_f32_f16): row dot products of 16-bit images
__device__ __forceinline__ float dot2_acc(const bf16x2 a, const bf16x2 b, const float c) {
#ifdef CARA_F16_OPERANDS
  return __builtin_amdgcn_fdot2(a, b, c, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
#endif
}

// cara_layernorm_bwd_ex with the running gradient dx_in read on every dx_in_every-th row only (0: every row).  Library-internal
// (norm_misc.hip -> vit.hip's last block), not part of include/cara_hip.h.
int cara_layernorm_bwd_rows_in(const void* dy, const float* x, long ldx, const float* gamma, const float* mean, const float* rstd,
                               const float* dx_in, float* dx_out, void* dyb, const float* rowscale, int rows_per_sample, int M, int C,
                               const void* Vst, int rank, int Rp, void* G, void* Gt, int ldt, int dyb_panels, int dx_in_every,
                               void* stream);

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

#define CARA_CHECK_LAUNCH()                                   \
  do {                                                        \
    if (hipGetLastError() != hipSuccess) return CARA_E_LAUNCH; \
  } while (0)

// async 16-byte global -> LDS copy: LDS destination = lds_wave_base + lane * 16 (lane-linear),
// global source is per lane.  Completion is tracked by vmcnt.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 output resolution): for y >= 0
//   erf(y) = 1 - P(t) exp(-y^2),  t = 1 / (1 + 0.3275911 y),  P = t (a1 + t (a2 + t (a3 + t (a4 + t a5))))
// one v_rcp + one v_exp + a handful of FMAs instead of the branchy libm erff.  The GELU epilogues are VALU-bound (every
// instruction per element costs ~1 us of an fc1 / fc2-dX product, tools: CARA_ABLATE_GELU), so the two forms below are
// written for instruction count: the argument scalings are folded into the constants, |u| is a source modifier, and
// the odd symmetry is used so that no compare / select is needed:
//   gelu(u)  = u Phi(u) = max(u, 0) - |u| (P/2) e          (u erf(u / sqrt 2) is even)
//   gelu'(u) = Phi(u) + u phi(u),  Phi(u) = 1/2 + copysign(1/2 - (P/2) e, u),  phi(u) = e / sqrt(2 pi)
// with e = exp(-u^2 / 2) shared by both terms of the derivative.
__device__ __forceinline__ float gelu_half_poly(float au, float& e, float usq) {
  // au = |u|; returns P(t) / 2 for y = au / sqrt(2) and e = exp(-u^2 / 2) = exp2(-u^2 log2(e) / 2)
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752f, au, 1.0f));
  e = __builtin_amdgcn_exp2f(usq * (-0.5f * 1.4426950408889634f));
  return t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f + t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
}
__device__ __forceinline__ float gelu_erf(float u) {
#ifdef CARA_ABLATE_GELU
  return u * 0.5f;   // timing diagnostic: what the erf arithmetic of the GELU epilogues costs
#endif
  float e;
  const float au = fabsf(u);
  const float hp = gelu_half_poly(au, e, u * u);
  return __builtin_fmaf(-(au * hp), e, fmaxf(u, 0.f));
}
__device__ __forceinline__ float gelu_erf_grad(float u) {
#ifdef CARA_ABLATE_GELU
  return u * 0.5f;
#endif
  float e;
  const float hp = gelu_half_poly(fabsf(u), e, u * u);
  const float half_erf = __builtin_fmaf(-hp, e, 0.5f);                    // erf(|u| / sqrt 2) / 2, in [0, 1/2]
  const float cdf = 0.5f + __builtin_copysignf(half_erf, u);
  return __builtin_fmaf(u * 0.39894228040143268f, e, cdf);
}

// gelu(u) and gelu'(u) together (the epilogue that rebuilds h = gelu(u) next to dH = dY gelu'(u)): the two share e and P;
// each result is bitwise what gelu_erf / gelu_erf_grad return
__device__ __forceinline__ void gelu_erf_both(float u, float& g, float& gp) {
#ifdef CARA_ABLATE_GELU
  g = gp = u * 0.5f;
  return;
#endif
  float e;
  const float au = fabsf(u);
  const float hp = gelu_half_poly(au, e, u * u);
  g = __builtin_fmaf(-(au * hp), e, fmaxf(u, 0.f));
  const float half_erf = __builtin_fmaf(-hp, e, 0.5f);
  gp = __builtin_fmaf(u * 0.39894228040143268f, e, 0.5f + __builtin_copysignf(half_erf, u));
}

// D[16 x 16] += A[16 x 16] B[16 x 16] on four operand values per lane (lane l: A[l % 16][4 (l / 16) ..], B[4 (l / 16) ..][l % 16]):
// the K = 16 MFMA, for products whose reduction index comes in runs of four (one transposing LDS read per fragment)
__device__ __forceinline__ f32x4 mfma_16x16x16(bf16x4 a, bf16x4 b, f32x4 c) {
#ifdef CARA_F16_OPERANDS
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#else
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous run of logical tiles (neighbouring tiles share operand panels
// and then hit that XCD's private L2).  Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// The same with the XCD's run permuted so that the workgroups that land on ONE CU hold consecutive logical tiles.  Observed
// placement (tools/micro/placement_map.hip, speed only, never correctness): inside an XCD the j-th workgroup goes to CU j % 32
// (4 shader engines x 8 CUs, round-robin), so while the whole grid is resident (<= 4 per CU: nwg <= 1024) workgroups j,
// j + 32, j + 64, j + 96 of an XCD share a CU.  With consecutive tiles of a row-major order they share their A row panel:
// the CU's L1 then serves two of three (three of four) requests for an A tile of a K step.
__device__ __forceinline__ int xcd_remap_cu(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  if (cnt > 128) return base + j;
  const int c = j & 31, sl = j >> 5, sf = cnt >> 5, rem = cnt & 31;
  return base + c * sf + (c < rem ? c : rem) + sl;
}
