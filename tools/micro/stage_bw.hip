// How fast can a CU stage an L2-resident operand tile into LDS?  (a) 16-byte global_load_lds (LDS-DMA),
// (b) global_load_dwordx4 -> VGPR -> ds_write_b128.  Each 256-thread workgroup stages `KB` KiB per iteration
// (double-buffered in LDS, one barrier per iteration like the GEMM K loop, but no MFMA), WPC workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o stage_bw tools/micro/stage_bw.hip && ./stage_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int KB>
__global__ __launch_bounds__(256) void stage_kernel(const char* __restrict__ src, unsigned* __restrict__ out, int iters, size_t span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BYTES = KB * 1024, PIECES = BYTES / (256 * 16);   // 16-B pieces per thread per iteration
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + ((size_t)blockIdx.x * 64 * 1024) % span;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    char* buf = smem + (it & 1) * BYTES;
    const char* s = base + ((size_t)(it & 3) * BYTES);
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int q = wave * PIECES + p;   // 1-KiB piece of this wave
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + q * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[PIECES];
#pragma unroll
      for (int p = 0; p < PIECES; ++p) v[p] = *reinterpret_cast<const u32x4*>(s + (wave * PIECES + p) * 1024 + lane * 16);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *reinterpret_cast<u32x4*>(buf + (wave * PIECES + p) * 1024 + lane * 16) = v[p];
    }
    __syncthreads();
    acc += *reinterpret_cast<unsigned*>(buf + ((tid * 16 + it * 4) & (BYTES - 1)));
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int MODE, int KB>
void run(const char* name, int wpc, const char* src, unsigned* out, size_t span) {
  const int cus = 256, iters = 2000;
  const size_t lds = 2 * KB * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<MODE, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((stage_kernel<MODE, KB>), dim3(cus * wpc), dim3(256), lds, 0, src, out, 50, span);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((stage_kernel<MODE, KB>), dim3(cus * wpc), dim3(256), lds, 0, src, out, iters, span);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)cus * wpc * iters * KB * 1024;
  printf("%-28s %2d KiB/iter x %d WG/CU: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, KB, wpc, bytes / (ms * 1e-3) / cus / 1e9,
         bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t span = 24u << 20;   // 24 MiB source: L2 + Infinity Cache resident
  char* src; unsigned* out;
  CK(hipMalloc(&src, span + (1 << 20))); CK(hipMemset(src, 1, span + (1 << 20))); CK(hipMalloc(&out, 1 << 20));
  for (int wpc : {1, 2, 4}) {
    run<0, 16>("LDS-DMA (global_load_lds)", wpc, src, out, span);
    run<1, 16>("VGPR + ds_write_b128", wpc, src, out, span);
  }
  for (int wpc : {1, 2}) {
    run<0, 32>("LDS-DMA (global_load_lds)", wpc, src, out, span);
    run<1, 32>("VGPR + ds_write_b128", wpc, src, out, span);
  }
  return 0;
}
