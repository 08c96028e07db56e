// Synthetic GEMM K loop, built up one component at a time, to see what a 128x128x32 step really costs on a CU
// with 4 resident workgroups: (S) stage 16 KiB by LDS-DMA with the GEMM's wait + barrier, (R) + the 8 fragment
// ds_read_b128 of a 64x64 wave tile, (M) + its 16 MFMAs.  L2-resident source, no epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o kloop_bw tools/micro/kloop_bw.hip && ./kloop_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) * 3)) << 4); }

// MODE bits: 1 = stage, 2 = fragment reads, 4 = MFMA; PF = K steps of prefetch distance (1 = the GEMM's double buffer,
// 2 = three-slot ring with a counted wait)
// MODE bit 8: GEMM-like source addressing -- operands are row-major [rows][768] bf16, a K step takes 64 bytes of each
// of 128 rows (16 half cache lines per wave instruction) instead of 1 KiB contiguous
template <int MODE, int PF>
__global__ __launch_bounds__(256, 4) void kloop(const char* __restrict__ src, float* __restrict__ out, int iters, size_t span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = 16384, NSLOT = PF + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* base = src + ((size_t)blockIdx.x * 64 * 1024) % span;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int it) {
    char* buf = smem + (it % NSLOT) * SLOT;
    const char* s = base + (size_t)(it & 3) * SLOT;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int q = wave * 4 + p;
      const char* g = s + q * 1024 + lane * 16;
      if (MODE & 8) {   // pieces 0-7: A rows, 8-15: B rows; row r of the tile, K step it: 64 B at r * 1536 + (it % 24) * 64
        const int r = (q & 7) * 16 + (lane >> 2);
        g = src + ((size_t)(blockIdx.x % 96) * 128 + (q >> 3) * 12288 + r) * 1536 + (it % 24) * 64 + (lane & 3) * 16;
        // MODE bit 16: the B operand (pieces 8-15) is K-panel major, [K/32][N = 3072][32]: a tile's K step is 8 KiB contiguous
        if ((MODE & 16) && q >= 8)
          g = src + (40u << 20) + ((size_t)(it % 24) * 3072 + (blockIdx.x % 24) * 128) * 64 + (q & 7) * 1024 + lane * 16;
        // MODE bit 32: the A operand K-panel major as well, [K/32][M = 12608][32]
        if ((MODE & 32) && q < 8)
          g = src + ((size_t)(it % 24) * 12608 + (blockIdx.x % 96) * 128) * 64 + (q & 7) * 1024 + lane * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, 0, 0);
    }
  };
  if (MODE & 1)
    for (int t = 0; t < PF; ++t) stage(t);
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
      if (PF == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // one younger K step (4 pieces) may stay in flight
    }
    __syncthreads();
    if (MODE & 1) stage(it + PF);
    const char* sA = smem + (it % NSLOT) * SLOT;
    const char* sB = sA + 8192;
    bf16x8 a[4], b[4];
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(sA + swz32(wr * 64 + i * 16 + fr, fq));
        b[i] = *reinterpret_cast<const bf16x8*>(sB + swz32(wc * 64 + i * 16 + fr, fq));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = bf16x8{}; b[i] = bf16x8{}; a[i][0] = (__bf16)(float)it; b[i][0] = (__bf16)1.f; }
    }
    if (MODE & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][0][0] += (float)a[i][0] + (float)b[i][1];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456f) out[blockIdx.x] = s;
}

// 256x128x32 step: 8 waves of 64x64, 24 KiB staged per step (3 pieces per wave), two workgroups per CU
__global__ __launch_bounds__(512, 4) void kloop256(const char* __restrict__ src, float* __restrict__ out, int iters, size_t span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = 24576;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* base = src + ((size_t)blockIdx.x * 96 * 1024) % span;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int it) {
    char* buf = smem + (it & 1) * SLOT;
    const char* s = base + (size_t)(it & 3) * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int q = wave * 3 + p;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + q * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, 0, 0);
    }
  };
  stage(0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage(it + 1);
    const char* sA = smem + (it & 1) * SLOT;
    const char* sB = sA + 16384;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const bf16x8*>(sA + swz32(wr * 64 + i * 16 + fr, fq));
      b[i] = *reinterpret_cast<const bf16x8*>(sB + swz32(wc * 64 + i * 16 + fr, fq));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456f) out[blockIdx.x] = s;
}

void run256(int wpc, const char* src, float* out, size_t span) {
  const int cus = 256, iters = 2000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kloop256), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kloop256, dim3(cus * wpc), dim3(512), 2 * 24576, 0, src, out, 50, span);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kloop256, dim3(cus * wpc), dim3(512), 2 * 24576, 0, src, out, iters, span);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = 2.0 * 256 * 128 * 32 * (double)cus * wpc * iters / (ms * 1e-3) / 1e12;
  printf("%-44s %d WG/CU: %6.3f us per K step per WG  -> %7.1f TF/s-equivalent\n", "the K loop at 256x128 (8 waves, 24 KiB/step)", wpc,
         ms * 1e3 / iters, tf);
}


// 128x128x64 step (BK = 64): 32 KiB per step, 8 pieces of 1 KiB per wave, 32 MFMAs per wave; double buffer = 64 KiB
// per workgroup.  SEG = 0: 1-KiB contiguous pieces; SEG = 1: GEMM-like, a piece is 8 rows x 128 B (full cache lines)
// of a row-major [rows][3072 B] operand.
template <int SEG>
__global__ __launch_bounds__(256, 2) void kloop64(const char* __restrict__ src, float* __restrict__ out, int iters, size_t span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = 32768;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* base = src + ((size_t)blockIdx.x * 128 * 1024) % span;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int it) {
    char* buf = smem + (it & 1) * SLOT;
    const char* s = base + (size_t)(it & 3) * SLOT;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = wave * 8 + p;   // pieces 0-15: A rows (8 rows each), 16-31: B rows
      const char* g = s + q * 1024 + lane * 16;
      if (SEG) {
        const int r = (q & 15) * 8 + (lane >> 3);
        g = src + ((size_t)(blockIdx.x % 96) * 128 + (q >> 4) * 12288 + r) * 3072 + (it % 24) * 128 + (lane & 7) * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, 0, 0);
    }
  };
  stage(0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage(it + 1);
    const char* sA = smem + (it & 1) * SLOT;
    const char* sB = sA + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(sA + (wr * 64 + i * 16 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
        b[i] = *reinterpret_cast<const bf16x8*>(sB + (wc * 64 + i * 16 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456f) out[blockIdx.x] = s;
}

template <int SEG>
void run64(int wpc, const char* src, float* out, size_t span) {
  const int cus = 256, iters = 1000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kloop64<SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kloop64<SEG>, dim3(cus * wpc), dim3(256), 2 * 32768, 0, src, out, 50, span);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kloop64<SEG>, dim3(cus * wpc), dim3(256), 2 * 32768, 0, src, out, iters, span);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = 2.0 * 128 * 128 * 64 * (double)cus * wpc * iters / (ms * 1e-3) / 1e12;
  printf("%-44s %d WG/CU: %6.3f us per K step per WG  -> %7.1f TF/s-equivalent\n",
         SEG ? "the K loop at 128x128x64, 128-B row segments" : "the K loop at 128x128x64, contiguous pieces", wpc, ms * 1e3 / iters, tf);
}

template <int MODE, int PF>
void run(const char* name, int wpc, const char* src, float* out, size_t span) {
  const int cus = 256, iters = 2000;
  const size_t lds = (PF + 1) * 16384;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kloop<MODE, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((kloop<MODE, PF>), dim3(cus * wpc), dim3(256), lds, 0, src, out, 50, span);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kloop<MODE, PF>), dim3(cus * wpc), dim3(256), lds, 0, src, out, iters, span);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us_per_step = ms * 1e3 / iters;
  const double tf = 2.0 * 128 * 128 * 32 * (double)cus * wpc * iters / (ms * 1e-3) / 1e12;
  printf("%-44s %d WG/CU: %6.3f us per K step per WG  -> %7.1f TF/s-equivalent\n", name, wpc, us_per_step, tf);
}

int main() {
  const size_t span = 96u << 20;
  char* src; float* out;
  CK(hipMalloc(&src, span + (1 << 20))); CK(hipMemset(src, 0x3c, span + (1 << 20))); CK(hipMalloc(&out, 1 << 20));
  for (int wpc : {1, 2}) { run64<0>(wpc, src, out, span); run64<1>(wpc, src, out, span); }
  for (int wpc : {3}) { run<7, 1>("stage + frag reads + MFMA (the K loop)", wpc, src, out, span); run<15, 1>("the K loop, GEMM-like 64-B row segments", wpc, src, out, span); }
  for (int wpc : {1, 2, 4}) {
    run<1, 1>("stage", wpc, src, out, span);
    run<3, 1>("stage + frag reads", wpc, src, out, span);
    run<6, 1>("frag reads + MFMA (no staging)", wpc, src, out, span);
    run<4, 1>("MFMA only", wpc, src, out, span);
    run<7, 1>("stage + frag reads + MFMA (the K loop)", wpc, src, out, span);
    run<7, 2>("the K loop, 2 K steps of prefetch", wpc, src, out, span);
    run<9, 1>("stage, GEMM-like 64-B row segments", wpc, src, out, span);
    run<15, 1>("the K loop, GEMM-like 64-B row segments", wpc, src, out, span);
    run<31, 1>("the K loop, A row segments + B K-panel major", wpc, src, out, span);
    run<63, 1>("the K loop, A and B K-panel major", wpc, src, out, span);
  }
  for (int wpc : {1, 2, 3}) run256(wpc, src, out, span);
  return 0;
}
