// Device-side helpers shared by the gfx950 kernels of libcara_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cara_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

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

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 output resolution):
// one v_rcp + one v_exp + 6 FMAs instead of the branchy libm erff.  e = exp(-x^2) is returned too:
// with x = u / sqrt(2) it is also the Gaussian of gelu'(u), so forward and backward share it.
__device__ __forceinline__ float erf_as(float x, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  e = __expf(-ax * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * e;
  return x < 0.f ? -r : r;
}
__device__ __forceinline__ float gelu_erf(float u) {
  float e;
  return 0.5f * u * (1.0f + erf_as(u * 0.70710678118654752f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float u) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_as(u * 0.70710678118654752f, e));
  return cdf + u * 0.39894228040143268f * e;
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
