// AdamW over the trainable tensors of a step (12 CP tensors + the classifier head: ~1.2e5 elements at rank 16) as ONE launch.
// The reference steps torch.optim.AdamW (image_classification/vit_cp.py:185, :50); its fused CUDA/HIP path spends 42 us per step
// on these few small tensors (rocprofv3: multi_tensor_apply_kernel, 20 workgroups), a kernel of our own 3-4 us.
//
// Arithmetic = torch.optim.AdamW (amsgrad = False, maximize = False), element by element in fp32:
//   p   *= 1 - lr * weight_decay                         (decoupled decay)
//   m    = m + (1 - beta1) (g - m)                        (exp_avg.lerp_)
//   v    = beta2 v + (1 - beta2) g g
//   p   -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The tensors come BY VALUE in the kernel arguments (pointers and sizes of up to 32 tensors, 1.3 KiB): no table in device
// memory, nothing to upload when a learning rate changes.
#include "common.h"

namespace {

constexpr int ADAMW_CHUNK = 1024;   // elements per workgroup (256 threads x 4)

__global__ __launch_bounds__(256) void adamw_kernel(const cara_adamw_args a) {
  // which tensor this workgroup works on: a prefix walk over <= 32 entries (uniform)
  int t = 0, first = 0;
  for (; t < a.ntensors; ++t) {
    const int nchunk = (int)((a.t[t].n + ADAMW_CHUNK - 1) / ADAMW_CHUNK);
    if ((int)blockIdx.x < first + nchunk) break;
    first += nchunk;
  }
  if (t >= a.ntensors) return;
  if (a.skip_flag && *a.skip_flag != 0.f) return;   // the step's gradients overflowed under the loss scale: nothing moves
  const cara_adamw_tensor& T = a.t[t];
  float lr = a.lr[T.group], bc1 = a.bias_correction1, bc2s = a.bias_correction2_sqrt;
  if (a.dyn) {   // the capturable form: step count and learning rates live in device memory (cara_adamw_args::dyn)
    const float tt = a.dyn[0];
    lr = a.dyn[1 + T.group];
    bc1 = 1.f - powf(1.f - a.one_minus_beta1, tt);
    bc2s = sqrtf(1.f - powf(a.beta2, tt));
  }
  const float wd = a.weight_decay[T.group];
  const float decay = 1.f - lr * wd;
  const float step_size = lr / bc1;
  float* __restrict__ p = T.p;
  const float* __restrict__ g = T.g;
  float* __restrict__ m = T.m;
  float* __restrict__ v = T.v;
  const size_t base = (size_t)((int)blockIdx.x - first) * ADAMW_CHUNK;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = base + (size_t)j * 256 + threadIdx.x;
    if (i >= T.n) continue;
    const float gi = g[i];
    float pi = p[i] * decay;
    const float mi = m[i] + a.one_minus_beta1 * (gi - m[i]);
    const float vi = a.beta2 * v[i] + a.one_minus_beta2 * gi * gi;
    const float denom = sqrtf(vi) / bc2s + a.eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// GradScaler's update rule on the device: one thread (cara_hip.h, cara_amp_update)
__global__ void amp_update_kernel(float* __restrict__ st, const float* __restrict__ found_inf, float growth, float backoff,
                                  int interval, float max_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*found_inf != 0.f) {
    st[0] = fmaxf(st[0] * backoff, 1.f);
    st[1] = 0.f;
    st[2] += 1.f;
  } else {
    const float n = st[1] + 1.f;
    if (n >= (float)interval) {
      st[0] = fminf(st[0] * growth, max_scale);
      st[1] = 0.f;
    } else {
      st[1] = n;
    }
  }
}

}  // namespace

extern "C" int cara_amp_update(float* state, const float* found_inf, float growth, float backoff, int interval, float max_scale,
                               void* stream) {
  if (!state || !found_inf || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) || interval <= 0 || !(max_scale >= 1.f)) return CARA_E_ARG;
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, found_inf, growth, backoff,
                     interval, max_scale);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_adamw_step(const cara_adamw_args* a, void* stream) {
  if (!a || a->ntensors <= 0 || a->ntensors > CARA_ADAMW_MAX_TENSORS || (a->step <= 0 && !a->dyn)) return CARA_E_ARG;
  size_t chunks = 0;
  for (int t = 0; t < a->ntensors; ++t) {
    const cara_adamw_tensor& T = a->t[t];
    if (!T.p || !T.g || !T.m || !T.v || T.n == 0 || T.group < 0 || T.group >= CARA_ADAMW_MAX_GROUPS) return CARA_E_ARG;
    chunks += (T.n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
  }
  if (chunks > 0x7fffffffu) return CARA_E_ARG;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)chunks), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
