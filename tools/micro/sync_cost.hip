// What does a cross-stream hand-off cost on the MAIN stream?  A chain of N ~20 us kernels on one stream with
//   (a) nothing between, (b) a hipEventRecord between, (c) record + side stream waits + side kernel + record join,
//   (d) as (c) plus the main stream waiting for the previous join,  each with several event flag sets.
// build: hipcc -O3 --offload-arch=gfx950 sync_cost.hip -o sync_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main() {
  const int N = 400, blocks = 1024, thr = 256;
  float *a, *b;
  hipMalloc(&a, blocks * thr * 4); hipMalloc(&b, blocks * thr * 4);
  hipMemset(a, 0, blocks * thr * 4); hipMemset(b, 0, blocks * thr * 4);
  hipStream_t m, s;
  hipStreamCreateWithFlags(&m, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const unsigned flagsets[3] = {hipEventDisableTiming, hipEventDisableTiming | hipEventReleaseToDevice, hipEventDefault};
  const char* fname[3] = {"disable_timing", "disable_timing|release_to_device", "default(timing)"};
  const int iters = 700;
  for (int fs = 0; fs < 1; ++fs) {
    std::vector<hipEvent_t> fork(N), join(N);
    for (int i = 0; i < N; ++i) { hipEventCreateWithFlags(&fork[i], flagsets[fs]); hipEventCreateWithFlags(&join[i], flagsets[fs]); }
    for (int mode = 0; mode < 7; ++mode) {
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(thr), 0, m, a, iters);
          if (mode == 1) hipEventRecord(fork[i], m);
          if (mode == 2 || mode == 3) {
            hipEventRecord(fork[i], m);
            hipStreamWaitEvent(s, fork[i], 0);
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(thr), 0, s, b, iters / 2);
            hipEventRecord(join[i], s);
            if (mode == 3 && i > 0) hipStreamWaitEvent(m, join[i - 1], 0);
          }
          if (mode == 5) { hipEventRecord(fork[i], m); hipStreamWaitEvent(s, fork[i], 0); hipEventRecord(join[i], s); }
          if (mode == 6) { hipEventRecord(fork[i], m); hipStreamWaitEvent(s, fork[i], 0); hipEventRecord(join[i], s); hipStreamWaitEvent(m, join[i], 0); }
          if (mode == 4 && (i % 4) == 3) {   // one fork + one join per four kernels
            hipEventRecord(fork[i], m);
            hipStreamWaitEvent(s, fork[i], 0);
            for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(thr), 0, s, b, iters / 2);
            hipEventRecord(join[i], s);
            if (i > 4) hipStreamWaitEvent(m, join[i - 4], 0);
          }
        }
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        if (us < best) best = us;
      }
      const char* mn[7] = {"kernels only", "+record", "+record, side wait+kernel+join record", "... + main waits prev join", "fork/join once per 4 kernels", "record, side wait, join record (no side kernel)", "... + main waits that join at once"};
      printf("%-34s %-44s %7.2f us per main kernel\n", fname[fs], mn[mode], best);
    }
  }
  // stream memory operations instead of events: main writes a counter, the side stream waits for it
  {
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint32_t* flag = nullptr;
    if (can && hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) == hipSuccess) {
      hipMemset(flag, 0, 8);
      for (int mode = 0; mode < 3; ++mode) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
          hipDeviceSynchronize();
          hipMemset(flag, 0, 8);
          hipDeviceSynchronize();
          auto t0 = std::chrono::steady_clock::now();
          for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(thr), 0, m, a, iters);
            if (mode >= 1) hipStreamWriteValue32(m, flag, (uint32_t)(i + 1), 0);
            if (mode == 2) {
              hipStreamWaitValue32(s, flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu);
              hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(thr), 0, s, b, iters / 2);
            }
          }
          hipDeviceSynchronize();
          const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
          if (us < best) best = us;
        }
        const char* mn[3] = {"kernels only", "+hipStreamWriteValue32", "+write, side hipStreamWaitValue32 + kernel"};
        printf("%-34s %-44s %7.2f us per main kernel\n", "stream memory ops", mn[mode], best);
      }
    }
  }
  return 0;
}
