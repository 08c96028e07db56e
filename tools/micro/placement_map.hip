// Which CU does workgroup b of a 1-D grid land on?  Diagnostic for the "L1 sharing by construction" idea of DESIGN.md 7.3:
// prints, per XCD, the CU (shader engine, CU id) of the first 160 XCD-local workgroups (b = xcd + 8 j) of a 594-block launch
// with the GEMM's resources (256 threads, 36 KiB LDS, ~100 VGPRs are not modelled), every block spinning ~20 us so that
// all of them are resident together.  Usage: hipcc --offload-arch=gfx950 -O2 placement_map.hip -o placement_map && ./placement_map [grid]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256) void where_kernel(unsigned* out, unsigned long long* t) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = xcc;
    t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}    // 20 us at 100 MHz
  if (threadIdx.x == 1) smem[0] = 1;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 594;
  const int lds = argc > 2 ? atoi(argv[2]) : 36864;
  unsigned* d;
  unsigned long long* dt;
  hipMalloc(&d, grid * 8);
  hipMalloc(&dt, grid * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), lds, 0, d, dt);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(grid * 2);
  std::vector<unsigned long long> ht(grid);
  hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), dt, grid * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < grid; ++b) tmin = ht[b] < tmin ? ht[b] : tmin;
  for (int x = 0; x < 8; ++x) {
    printf("blocks b = %d + 8 j (xcc of block: %u):", x, h[x * 2 + 1] & 0xf);
    for (int j = 0; x + 8 * j < grid && j < 80; ++j) {
      const unsigned hw = h[(x + 8 * j) * 2];
      printf(" %u.%u.%02u", (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf);
    }
    printf("\n");
  }
  // how many distinct XCDs does each residue class b % 8 see, and start-time spread
  int bad = 0;
  for (int b = 0; b < grid; ++b) bad += (h[b * 2 + 1] & 0xf) != (h[(b % 8) * 2 + 1] & 0xf);
  unsigned long long tmax = 0;
  for (int b = 0; b < grid; ++b) tmax = ht[b] > tmax ? ht[b] : tmax;
  printf("blocks whose XCD differs from that of block b %% 8: %d; start spread %.2f us\n", bad, (tmax - tmin) / 100.0);
  return 0;
}
