// Library identification for libcara_hip.so.
#include <dlfcn.h>

#include "common.h"

extern "C" int cara_abi_version(void) { return 14; }
extern "C" const char* cara_build_arch(void) { return "gfx950"; }
extern "C" const char* cara_operand_type(void) { return CARA_OPERAND_TYPE; }

extern "C" size_t cara_sizeof_struct(int which) {
  switch (which) {
    case CARA_STRUCT_GEMM_ARGS: return sizeof(cara_gemm_args);
    case CARA_STRUCT_GEOM: return sizeof(cara_geom);
    case CARA_STRUCT_CP: return sizeof(cara_cp);
    case CARA_STRUCT_PACK_LAYOUT: return sizeof(cara_pack_layout);
    case CARA_STRUCT_LAYER_GRADS: return sizeof(cara_layer_grads);
    case CARA_STRUCT_VIT_WEIGHTS: return sizeof(cara_vit_weights);
    case CARA_STRUCT_VIT_SHAPE: return sizeof(cara_vit_shape);
    case CARA_STRUCT_TS_REDUCE: return sizeof(cara_ts_reduce);
    case CARA_STRUCT_LINEAR: return sizeof(cara_linear);
    case CARA_STRUCT_ADAMW_ARGS: return sizeof(cara_adamw_args);
    default: return 0;
  }
}
extern "C" size_t cara_sizeof_gemm_args(void) { return sizeof(cara_gemm_args); }

// Diagnostic: one ds_read_b64_tr_b16 per lane over an LDS image sm[i] = i (16-bit), with the byte
// address of every lane given by the caller.  Used by tests to pin the lane semantics of the
// transposing LDS read that the attention kernels rely on.
namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__global__ void tr_probe_kernel(const int* __restrict__ byte_addr, short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (short)i;
  __syncthreads();
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4_t*)(reinterpret_cast<char*>(sm) + byte_addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
}  // namespace

extern "C" int cara_debug_tr_probe(const int* byte_addr, short* out, void* stream) {
  if (!byte_addr || !out) return CARA_E_ARG;
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), byte_addr, out);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// The optional RCCL wrapper of SURVEY.md 8(b): the step's ONE collective -- a SUM all-reduce of the flat fp32 gradient buffer
// (487 696 bytes at the headline configuration: 121 923 gradients + the found-inf word) -- for a host that owns an ncclComm_t
// (a C++ trainer; the Python side of this repository goes through torch.distributed, whose communicator is not exposed).
// librccl is looked up at the first call (dlopen), so the library loads and runs on one GPU without it.
extern "C" int cara_allreduce_flat(void* nccl_comm, float* buf, size_t count, void* stream) {
  typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  static const allreduce_fn fn = [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return h ? reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce")) : nullptr;
  }();
  if (!nccl_comm || !buf || !count) return CARA_E_ARG;
  if (!fn) return CARA_E_LAUNCH;   // no RCCL in this process's library path
  constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // (rccl.h: ncclDataType_t / ncclRedOp_t)
  return fn(buf, buf, count, kNcclFloat32, kNcclSum, nccl_comm, static_cast<hipStream_t>(stream)) == 0 ? CARA_OK : CARA_E_LAUNCH;
}
