// Which block indices share a CU?  1024 workgroups (256 threads, 32 KiB LDS: 4 per CU) all resident; prints, for the first
// few CUs, the block indices they hold, and the distribution of index differences between co-resident blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>
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
  const int n = 1024;
  CK(hipMalloc(&ids, n * 8)); CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
  hipLaunchKernelGGL(where, dim3(n), dim3(256), 32 * 1024, 0, ids, counter, n);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h(n * 2);
  CK(hipMemcpy(h.data(), ids, n * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, std::vector<int>> cu;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    cu[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)].push_back(i);
  }
  int shown = 0;
  std::map<int, int> diffs;
  for (auto& kv : cu) {
    auto v = kv.second; std::sort(v.begin(), v.end());
    if (shown++ < 12) { printf("cu %06x:", kv.first); for (int b : v) printf(" %d", b); printf("\n"); }
    for (size_t k = 1; k < v.size(); ++k) diffs[v[k] - v[k - 1]]++;
  }
  printf("differences between consecutive co-resident block indices:");
  for (auto& d : diffs) printf("  %d x%d", d.first, d.second);
  printf("\n");
  return 0;
}
