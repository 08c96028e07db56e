// Where do the workgroups of a partly filling grid land?  256-thread workgroups with 32 KiB of LDS (up to 4-5 fit a
// CU); each records (XCC, SE, CU) from the hardware id registers while all of them are resident (they spin until
// every workgroup has arrived, grid <= resident capacity).  Prints the histogram of workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 4) void where(unsigned* __restrict__ ids, unsigned* __restrict__ counter, int n) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ids[blockIdx.x * 2] = hw;
    ids[blockIdx.x * 2 + 1] = xcc;
    atomicAdd(counter, 1u);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n && spins < 2000000) { __builtin_amdgcn_s_sleep(8); ++spins; }
  }
  smem[threadIdx.x] = 0;
  __syncthreads();
}

int main() {
  unsigned *ids, *counter;
  CK(hipMalloc(&ids, 8192 * 8)); CK(hipMalloc(&counter, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int n : {256, 300, 512, 594, 768, 1024}) {
    CK(hipMemset(counter, 0, 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 32 * 1024, 0, ids, counter, n);
    CK(hipDeviceSynchronize());
    unsigned* h = (unsigned*)malloc(n * 8);
    CK(hipMemcpy(h, ids, n * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu;
    for (int i = 0; i < n; ++i) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
      per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    int hist[16] = {0};
    for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
    printf("grid %4d: %3zu CUs used; CUs holding k workgroups:", n, per_cu.size());
    for (int k = 1; k < 10; ++k) if (hist[k]) printf("  k=%d: %d", k, hist[k]);
    printf("\n");
    free(h);
  }
  return 0;
}
