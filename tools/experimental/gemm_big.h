// The 208 x 256 tile of the bf16 GEMM (gfx950): ONE 512-thread workgroup per CU.  Included by gemm.hip inside its
// anonymous namespace (uses swz32, glds16, epilogue_rows, the STAMP macros).
//
// Why a second tile.  In-kernel time stamps (tools/gemm_stamps.py) show what bounds the K loop of the 128 x 128 kernel:
// a CU takes in ~33-38 bytes per clock through its vector-memory path (L2 -> L1 -> LDS; the L1's outstanding requests
// x line size / L2 latency), whatever the number of resident workgroups or the prefetch depth: 4 workgroups x 16 KiB per
// 0.92 us, 3 x 24 KiB per 1.18 us, 2.3 x 18 KiB per 0.58 us -- always ~70 GB/s per CU.  A product's K-loop time is
// therefore (bytes staged by its busiest CU) / 70 GB/s.  The N = 768 products (594 tiles of 128 x 128: three on the
// busiest CUs, 96 K steps each) stage 4.7 MB there = 65 us of a 83 us launch.  This tile stages (208 + 256) x 64 B per K
// step for 3.25x the flops of a 128 x 128 step (0.55x the bytes per flop), and its 61 x 3 = 183 tiles (M = 12608) are one
// per CU: 2.85 MB on every CU that has one.
//
//  * 8 waves as 2 (M: row tiles 0..6 | 7..12) x 4 (N: 64 columns each); a wave of row 0 owns 7 x 4 accumulators of
//    v_mfma_f32_16x16x32_bf16, a wave of row 1 owns 6 x 4; waves w and w + 4 share a SIMD, so every SIMD issues 52 MFMAs
//    per K step.
//  * The two wave rows run half a K step apart (two barriers per step): while one row issues its MFMAs the other
//    issues its share of the LDS-DMA for step k + 2, reads its fragments of step k and waits -- the matrix pipe of
//    every SIMD always has one wave feeding it (MI355X_MICROARCH.md, two waves per SIMD).
//  * K steps of 32 in a ring of four 29-KiB slots (A 208 x 64 B | B 256 x 64 B, 16-byte-chunk XOR swizzle applied on
//    the global source address as in the 128 x 128 kernel), three steps in flight behind a counted vmcnt; the K-extension
//    ([T | Vs], Rp = 32) is one more step of the same loop with other base pointers.
//  * Row tiles start every `stride` <= 208 rows (stride = ceil(M / ceil(M / 208))): a tile computes 208 rows and
//    stores the first `stride` of them, so that M = 12608 is 61 equal tiles instead of 60 and a sliver.
#pragma once
#include <type_traits>

constexpr int GB_RT = 13;
constexpr int GB_TM = GB_RT * 16;                  // 208
constexpr int GB_TN = 256;
constexpr int GB_A_BYTES = GB_TM * 64;             // 13312
constexpr int GB_B_BYTES = GB_TN * 64;             // 16384
constexpr int GB_SLOT = GB_A_BYTES + GB_B_BYTES;   // 29696
constexpr int GB_SLOTS = 4;
constexpr int GB_LDS = GB_SLOTS * GB_SLOT;         // 89088
constexpr int GB_PIECES = GB_RT + GB_TN / 16;      // 29 one-KiB LDS-DMA pieces per K step
constexpr int GB_R0 = 7;                           // row tiles of wave row 0

constexpr int GB_LOADERS = 4;                      // DMA-only waves (waves 8..11)
constexpr int GB_THREADS = (8 + GB_LOADERS) * 64;  // 768

// Loader wave l issues the pieces l, l + 4, ..., (l + 28): eight of them for l = 0, seven for the others.  Pieces 0..12 are
// A's 16-row groups, 13..28 B's.  Per lane: the byte offset of its 16 bytes in a main K step (m) and in the extension step (e).
struct GbStage {
  unsigned m[8], e[8];
  int dst[8];          // LDS byte offset of the piece inside a slot (wave-uniform)
  int na;              // this wave's first `na` pieces are A pieces (4 for l = 0, else 3)
  bool eight;
};

__device__ __forceinline__ void gb_piece(const cara_gemm_args& p, const bool packed, const int m0, const int n0, const int idx, const int lane,
                                         unsigned& om, unsigned& oe, int& dst) {
  const int cg = (lane & 3) ^ (((lane >> 5) & 1) * 3);   // rows of a piece start at a multiple of 16: (row >> 3) & 1 = lane >> 5
  const bool isA = idx < GB_RT;
  const int row = (isA ? idx : idx - GB_RT) * 16 + (lane >> 2);
  dst = isA ? idx * 1024 : GB_A_BYTES + (idx - GB_RT) * 1024;
  int g = (isA ? m0 : n0) + row;
  const int gmax = (isA ? p.M : p.N) - 1;
  g = g < gmax ? g : gmax;
  const unsigned ld2 = isA ? (p.a_panels ? 64u : (unsigned)p.lda * 2u) : (packed ? 64u : (unsigned)p.ldb * 2u);
  om = (unsigned)g * ld2 + (unsigned)(cg * 16);
  oe = (unsigned)g * 64u + (unsigned)(cg * 16);
}

template <int MI>
__device__ __forceinline__ void gb_read(bf16x8 (&a)[GB_R0], bf16x8 (&b)[4], const char* slot, const int row0, const int wc, const int fr, const int fq) {
#pragma unroll
  for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(slot + swz32(row0 + i * 16 + fr, fq));
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(slot + GB_A_BYTES + swz32(wc * 64 + j * 16 + fr, fq));
}
template <int MI>
__device__ __forceinline__ void gb_mma(f32x4 (&acc)[GB_R0][4], const bf16x8 (&a)[GB_R0], const bf16x8 (&b)[4]) {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
}

// (loader waves) all of this wave's LDS-DMA but its `younger` youngest batches (the steps after the one about to be read) has landed
__device__ __forceinline__ void gb_wait_batch(const int younger, const bool eight) {
  static_assert(GB_SLOTS == 4, "vmcnt literals: two batches may stay in flight");
  if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (younger == 1) {
    if (eight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    if (eight) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  }
}
__device__ __forceinline__ void gb_barrier() {
  // pinned: hipcc moves register-only instructions (the MFMAs) across a bare inline-asm barrier
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void gb_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one workgroup's tile; `block` = its index among the nwg tiles
template <int EPI>
__device__ __forceinline__ void gemm_big_body(const cara_gemm_args& p, const int tiles_n, const int nwg, const int stride, const int block,
                                              char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  STAMP(0);
  const int tile = xcd_remap(block, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * stride, n0 = tn * GB_TN;
  const int nk = p.K >> 5;
  const int ntot = nk + (p.Rp ? 1 : 0);
  constexpr int D = GB_SLOTS - 1;   // steps 0 .. D - 1 are in flight before the loop; iteration k issues step k + D into the slot of step k - 1
  if (wave >= 8) {
    // ---- loader wave: nothing but LDS-DMA.  The MFMA waves never stall on a DMA issue, and the CU has 3..4 pieces per loader and
    // step queued at its vector-memory path at all times (its throughput grows with the number of waves that feed it) ----
    const int l = wave - 8;
    const bool packed = p.Bp != nullptr;
    unsigned om[8], oe[8];
    int dst[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) gb_piece(p, packed, m0, n0, t * GB_LOADERS + l < GB_PIECES ? t * GB_LOADERS + l : l, lane, om[t], oe[t], dst[t]);
    const bool eight = l + 7 * GB_LOADERS < GB_PIECES;
    const int na = l + 3 * GB_LOADERS < GB_RT ? 4 : 3;
    const char* A = static_cast<const char*>(p.A);
    const char* B = static_cast<const char*>(packed ? p.Bp : p.B);
    const long ksA = p.a_panels ? (long)p.a_panels * 64 : 64;   // bytes per K step
    const long ksB = packed ? (long)p.N * 64 : 64;
    const char* A2 = static_cast<const char*>(p.A2);
    const char* B2 = static_cast<const char*>(p.B2);
    // (constant indices and no select between the two offset arrays: both stay in registers)
    auto issue = [&](int s, int slot) {
      char* d = smem + slot * GB_SLOT;
      if (s < nk) {
        const char* pa = A + s * ksA;
        const char* pb = B + s * ksB;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < 7 || eight) glds16((t < 3 || (t == 3 && na == 4) ? pa : pb) + om[t], d + dst[t]);
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < 7 || eight) glds16((t < 3 || (t == 3 && na == 4) ? A2 : B2) + oe[t], d + dst[t]);
      }
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < ntot) issue(s, s);
    int nxt = D;
    for (int k = 0; k < ntot; ++k) {
      gb_wait_batch(ntot - 1 - k, eight);
      gb_barrier();                                   // A(k): step k is in LDS; the slot of step k - 1 is free
      if (k + D < ntot) issue(k + D, nxt);
      nxt = nxt == GB_SLOTS - 1 ? 0 : nxt + 1;
      gb_barrier();                                   // B(k)
    }
    __syncthreads();
    return;
  }
  const int wr = wave >> 2, wc = wave & 3;
  f32x4 acc[GB_R0][4];
#pragma unroll
  for (int i = 0; i < GB_R0; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[GB_R0], b[4];
  int cur = 0;
  if (wr == 0) {
    for (int k = 0; k < ntot; ++k) {
      gb_barrier();                                   // A(k)
      gb_read<GB_R0>(a, b, smem + cur * GB_SLOT, 0, wc, fr, fq);
      gb_lgkm0();
      gb_barrier();                                   // B(k)
      gb_mma<GB_R0>(acc, a, b);
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == GB_SLOTS - 1 ? 0 : cur + 1;
    }
  } else {
    for (int k = 0; k < ntot; ++k) {
      gb_lgkm0();                                     // this wave's reads of slot k - 1 are done before a loader refills it
      gb_barrier();                                   // A(k)
      if (k > 0) gb_mma<GB_RT - GB_R0>(acc, a, b);    // step k - 1
      __builtin_amdgcn_sched_barrier(0);
      gb_barrier();                                   // B(k)
      gb_read<GB_RT - GB_R0>(a, b, smem + cur * GB_SLOT, GB_R0 * 16, wc, fr, fq);
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == GB_SLOTS - 1 ? 0 : cur + 1;
    }
    gb_lgkm0();
    gb_mma<GB_RT - GB_R0>(acc, a, b);
  }
  __syncthreads();   // every wave is done with the ring
  STAMP(1);
  // epilogue.  Rows beyond this tile's share (m0 + stride) are not stored.  The row tiles of a wave that lie wholly inside the
  // share take the interior paths of gemm_epilogue.h (bf16 outputs converted in the accumulator layout; the input operand of
  // the residual / gelu' epilogues requested a pass group ahead); the rest -- the 15-row sliver at the end of a 207-row share,
  // edge tiles -- go 16 rows at a time through a wave-private [16][64] fp32 image.
  cara_gemm_args pm = p;
  pm.M = m0 + stride < p.M ? m0 + stride : p.M;
  constexpr int WAVE_STG = EPI_FAST_WAVE_BYTES > 16 * 64 * 4 ? EPI_FAST_WAVE_BYTES : 16 * 64 * 4;
  char* wstg = smem + wave * WAVE_STG;
  float* stg = reinterpret_cast<float*>(wstg);
  const int nrt = wr == 0 ? GB_R0 : GB_RT - GB_R0;
  const int mrow0 = m0 + (wr == 0 ? 0 : GB_R0 * 16);
  const int nw = n0 + wc * 64;
  int done = 0;   // row tiles already stored by an interior path
  const bool aligned = (p.ldc & 7) == 0 && (!p.bias || (nw & 3) == 0);
  auto interior = [&](auto nt_tag) {
    constexpr int NTI = decltype(nt_tag)::value;
    if constexpr (EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU) {
      epilogue_fast_bf16_rt<EPI, NTI>(pm, *reinterpret_cast<const f32x4(*)[NTI][4]>(&acc[0]), wstg, mrow0, nw, lane, 0);
      done = NTI;
    } else if constexpr (EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU) {
      if (EPI == CARA_EPI_DGELU || !p.rowscale || p.rows_per_sample >= NTI * 16) {
        epilogue_interior_aux<EPI, NTI, 1>(pm, *reinterpret_cast<const f32x4(*)[NTI][4]>(&acc[0]), stg, mrow0, nw, lane);
        done = NTI;
      }
    }
  };
  if (aligned) {   // (wave-uniform)
    if (wr == 0) {
      if (mrow0 + GB_R0 * 16 <= pm.M) interior(std::integral_constant<int, GB_R0>{});
    } else {
      if (mrow0 + (GB_RT - GB_R0) * 16 <= pm.M) interior(std::integral_constant<int, GB_RT - GB_R0>{});
      else if (mrow0 + (GB_RT - GB_R0 - 1) * 16 <= pm.M) interior(std::integral_constant<int, GB_RT - GB_R0 - 1>{});
    }
  }
#pragma unroll
  for (int i = 0; i < GB_R0; ++i) {
    if (i >= done && i < nrt && mrow0 + i * 16 < pm.M) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * 64 + j * 16 + fr] = acc[i][j][r];
      asm volatile("" ::: "memory");
      epilogue_rows<EPI, 16>(pm, stg, mrow0 + i * 16, nw, lane, 0);
      asm volatile("" ::: "memory");
    }
  }
  STAMP_END();
}

template <int EPI>
__global__ __launch_bounds__(GB_THREADS) void gemm_big_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_big_body<EPI>(p, tiles_n, nwg, stride, blockIdx.x, smem);
}

// The same tile carrying the two transposed skinny products of its linear (cara_gemm_with_tskinny): the blocks behind the GEMM's
// tiles are three 256-thread tskinny blocks each.  They land on the CUs the tiles leave free (M = 12608, N = 768: 183 tiles
// on 256 CUs) and run under the GEMM.
template <int EPI, bool COLSUM>
__global__ __launch_bounds__(GB_THREADS) void gemm_big_ts_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int stride,
                                                                 const TsProblem t0, const TsProblem t1, const int ldg, const int Mts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (b >= nwg) {
    constexpr int TSB = TsRing<2, 1>::BLOCK_BYTES;
    static_assert(3 * TSB <= GB_LDS, "three tskinny blocks share the tile's LDS");
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int nts = t0.nblk + t1.nblk;
    const int blk = 3 * (b - nwg) + sub;
    tskinny_body<2, COLSUM, 1>(t0, t1, ldg, Mts, blk < nts ? blk : nts - 1, smem + sub * TSB, threadIdx.x & 255, blk < nts);
  } else {
    gemm_big_body<EPI>(p, tiles_n, nwg, stride, b, smem);
  }
}

// what the tile takes: full 256-column strips, the plain K-extension, one product per launch
static bool big_tile_ok(const cara_gemm_args* a) {
  return (a->N % GB_TN) == 0 && a->M >= 1024 && (a->K % 32) == 0 && (a->Rp == 0 || (a->Rp == 32 && a->A2)) && a->batch <= 1 && !a->B3 &&
         !a->Ut && !a->c_panels;
}

struct TsPair;
template <int EPI>
int launch_big(const cara_gemm_args* a, hipStream_t st, const TsPair* ts);
