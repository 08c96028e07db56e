// bf16 MFMA GEMM with a rank-R K-extension and fused epilogues (gfx950 / MI355X only).
//
//   C = A[M,K] . B[N,K]^T  (+ A2[M,Rp] . B2[N,Rp]^T)  -> epilogue
//
// This one template serves every dense product of the adapted ViT block
// (/root/reference/src/cara/cara.py:25 qkv, :50 proj, :75 fc1, :87 fc2 and their dX backward):
// the CaRA delta  s * ((X U) (.) g) V^T  rides along as Rp extra columns of K
// ([X | T] . [W | Vs]^T), i.e. Rp/K extra MFMA work instead of the reference's
// second dense GEMM on a materialised dW.
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 =
// 4x4 v_mfma_f32_16x16x32_bf16 accumulators) -- 160x128 (80x64 per wave) for the N >= 3072 products, whose tiles are
// then 1.85 rounds of the workgroup slots instead of 2.32 --, BK = 32, two LDS slots of A rows x 32 + B 128x32 bf16
// (32 / 36 KiB -> FOUR workgroups per CU: one's prologue / epilogue runs under the K loops of the others),
// tiles staged by 16-byte global_load_lds (LDS-DMA, no VGPR round trip) issued one K-step ahead
// of the MFMAs that consume them, the K-extension step ([T | Vs]) included; one barrier per K-step.  LDS image is
// [row][32 bf16] with the 16-byte chunk index XOR-swizzled, applied on the global SOURCE address (the DMA destination is
// lane-linear) and again on the ds_read_b128 address: conflict-free fragment reads.
// 1-D grid remapped so that each XCD's L2 sees a contiguous run of tiles.
// What bounds it: a CU takes in ~70 GB/s through its vector-memory path whatever the number of
// resident workgroups or the prefetch depth; epilogues: gemm_epilogue.h.
// (Other structures that were built and measured slower on this model's shapes -- a 64-deep double buffer,
// 256x256 / 128x256 LDS-ring kernels, a persistent stream-K kernel, 256x128 tiles, a 208x256 one-workgroup-per-CU
// tile with DMA-only loader waves -- live in tools/experimental/, outside the library.)
// Also here: the adapter-inside form (gemm32ft_*: T = A Ut^T accumulated on the tiles the workgroup streams anyway, Rp = 32 or
// 64), the forms that carry a pair of transposed skinny products behind their tiles (gemm32_ts_kernel, gemm32ft_ts_kernel),
// the two-B-operand form of the exact weight-dropout mode (one K loop, both B tiles on the same A fragments), 8-wave
// 192 / 256-row tiles (a switch: measured a tie), and gemm_tn_kernel (C = At^T Bt from row-major operands through
// transposing LDS reads: the dense dW of the exact mode).
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "tskinny_body.h"
#include "gemm8.h"

#ifdef CARA_GEMM_STAMPS
// Diagnostic build (tools/gemm_stamps.py): wave 0 of every workgroup records s_memrealtime (100 MHz) at its start, at the
// end of its K loop and after its epilogue's stores have retired, plus where it ran, into a buffer of their own.
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int cara_debug_gemm_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define STAMP(slot)                                                                         \
  do {                                                                                      \
    if (g_stamp_buf && threadIdx.x == 0) g_stamp_buf[(size_t)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define STAMP_END()                                                                         \
  do {                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
    if (g_stamp_buf && threadIdx.x == 0) {                                                  \
      g_stamp_buf[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();           \
      g_stamp_buf[(size_t)blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    }                                                                                       \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_END() do {} while (0)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;   // (K granule the C ABI promises: K % 64 == 0)

constexpr int BK32 = 32;
constexpr int B32_BYTES = BN * BK32 * 2;        // 8 KiB: the B tile (128 rows x 64 B)

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) * 3)) << 4); }

// A ROWS x 64 B operand tile is ROWS/16 one-KiB LDS-DMA pieces (16 rows each), spread over the waves, with the
// row part of the address hoisted out of the K loop: off[t] = (clamped row) * ld * 2 + swizzled
// chunk * 16 is a per-thread 32-bit byte offset computed once per tile, and a K step only adds the wave-uniform
// k0 * 2 to the operand's base pointer (SGPR base + VGPR offset addressing: no vector arithmetic at all per piece).
// rocprofv3 counted 1 460 VALU instructions per wave and tile in the bf16-epilogue kernel (3.6 per MFMA), about
// two thirds of them this address arithmetic.  Needs the operand to span < 4 GiB (checked at dispatch).
template <int ROWS, int NW = 4>
struct TileOfs {
  static constexpr int NP = ROWS / 16;                  // one-KiB pieces of the tile
  static constexpr int PPW = (NP + NW - 1) / NW;         // pieces per wave (the last one only on the first NP % NW waves)
  static constexpr bool EVEN = NP % NW == 0;
  unsigned off[PPW];
  // piece t of a wave: blocked (wave * PPW + t) when the pieces divide evenly, interleaved (wave + NW * t) otherwise
  static __device__ __forceinline__ int piece(int wave, int t) { return EVEN ? wave * PPW + t : wave + NW * t; }
};
template <int ROWS, int NW = 4>
__device__ __forceinline__ TileOfs<ROWS, NW> tile_ofs(int ld, int r0, int rmax, int wave, int lane) {
  using T = TileOfs<ROWS, NW>;
  T o;
#pragma unroll
  for (int t = 0; t < T::PPW; ++t) {
    const int pc = T::piece(wave, t) < T::NP ? T::piece(wave, t) : T::NP - 1;
    const int r = pc * 16 + (lane >> 2);
    const int cg = (lane & 3) ^ (((r >> 3) & 1) * 3);
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;
    o.off[t] = (unsigned)gr * (unsigned)(ld * 2) + (unsigned)(cg * 16);
  }
  return o;
}
template <int ROWS, int NW = 4>
__device__ __forceinline__ void stage_tile32_pre(const bf16* __restrict__ P, int k0, const TileOfs<ROWS, NW>& o, char* lds_tile, int wave) {
  using T = TileOfs<ROWS, NW>;
  const char* base = reinterpret_cast<const char*>(P + k0);   // wave-uniform
#pragma unroll
  for (int t = 0; t < T::PPW; ++t)
    if (T::EVEN || T::piece(wave, t) < T::NP) glds16(base + o.off[t], lds_tile + T::piece(wave, t) * 1024);
}

template <int MI>
__device__ __forceinline__ void mma_tile32(const char* sA, const char* sB, f32x4 (&acc)[MI][4], int wr, int wc, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 a[MI], b[4];
#pragma unroll
  for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sA + swz32(wr * (MI * 16) + i * 16 + fr, fq));
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sB + swz32(wc * 64 + j * 16 + fr, fq));
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
}

// the same over TWO B tiles that share the A fragments: acc += A B^T + A B3^T (cara_gemm_args::B3)
template <int MI>
__device__ __forceinline__ void mma_tile32_2b(const char* sA, const char* sB, const char* sB3, f32x4 (&acc)[MI][4], int wr, int wc, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 a[MI], b[4], b3[4];
#pragma unroll
  for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sA + swz32(wr * (MI * 16) + i * 16 + fr, fq));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    b[j] = *reinterpret_cast<const bf16x8*>(sB + swz32(wc * 64 + j * 16 + fr, fq));
    b3[j] = *reinterpret_cast<const bf16x8*>(sB3 + swz32(wc * 64 + j * 16 + fr, fq));
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b3[j], acc[i][j], 0, 0, 0);
    }
}

// extension operands [rows, Rp] into 64-byte-row images, one image per 32 columns of Rp
template <int ROWS, int NW = 4>
__device__ __forceinline__ void stage_ext32(const bf16* __restrict__ P, int Rp, int r0, int rmax, int kk, char* lds_tile, int tid) {
  for (int idx = tid; idx < ROWS * 4; idx += NW * 64) {
    const int r = idx >> 2, c = idx & 3;
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;
    *reinterpret_cast<uint4*>(lds_tile + swz32(r, c)) = *reinterpret_cast<const uint4*>(P + (size_t)gr * Rp + kk * 32 + c * 8);
  }
}

// MI = 16-row MFMA tiles per wave along M: 4 -> 128x128 tile (32 KiB LDS, 4 workgroups/CU),
// 2 -> 64x128 tile (24 KiB LDS, 6 workgroups/CU, twice the tiles: used for the N = 768 products)
// NW = waves per workgroup (NW/2 along M x 2 along N): 4, or 8 with MI = 2 for a 128-row tile made of
// 32x64 wave tiles (more resident waves per CU)
// one workgroup's tile; `block` = its index among the nwg tiles of the product, `zb` = product index of a batched launch
// TWOB: every K step stages B AND B3 (cara_gemm_args::B3: same shape and ldb) next to the A tile and runs both products on
// the same A fragments (24 KiB slots: three workgroups per CU; the first form ran the K loop twice and staged A twice)
// ER: CARA_EPI_MULH with epilogue riders (cara_gemm_args::er_*, gemm_epilogue.h)
template <int EPI, int MI, int NW, bool TWOB = false, bool ER = false>
__device__ __forceinline__ void gemm32_body(const cara_gemm_args& p, const int tiles_n, const int nwg, const int gm,
                                            const int block, const size_t zb_in, char* smem) {
  constexpr int TBM = MI * 16 * (NW / 2);
  constexpr int A_BYTES = TBM * BK32 * 2;
  constexpr int SLOT = A_BYTES + (TWOB ? 2 : 1) * B32_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  STAMP(0);
  // Tile order: each XCD gets a contiguous run of logical tile indices (xcd_remap); inside the run
  // the tiles are walked in groups of `gm` tile rows, row index fastest (a "supertile"), so that the
  // ~128 tiles an XCD keeps in flight touch about sqrt(128)+sqrt(128) operand panels instead of
  // 5 + tiles_n -- rocprofv3 FETCH_SIZE showed 9x re-fetch on the N = 3072 products with the plain
  // row-major order (their W panels alone exceed the XCD's 4 MiB L2).
  const int tile = gm < 0 ? xcd_remap_cu(block, nwg) : xcd_remap(block, nwg);   // (gm < 0: CARA_GEMM_CUSHARE, plain order)
  int tm, tn;
  if (gm <= 1) {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  } else {
    const int tiles_m = nwg / tiles_n;
    const int per_group = gm * tiles_n;
    const int gid = tile / per_group, rem = tile - gid * per_group;
    const int first = gid * gm;
    const int gsz = (tiles_m - first) < gm ? (tiles_m - first) : gm;
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  }
  const int m0 = tm * TBM, n0 = tn * BN;
  // batched launch (blockIdx.y = product index): advance the operand pointers and the output offset
  const size_t zb = p.batch > 1 ? zb_in : 0;
  const bf16* __restrict__ A = static_cast<const bf16*>(p.A) + zb * p.strideA;
  // B from its K-panel-major image when there is one: row stride 64 B inside a panel, N * 32 elements per K step
  const bool packed = p.Bp != nullptr;
  const bf16* __restrict__ B = packed ? static_cast<const bf16*>(p.Bp) : static_cast<const bf16*>(p.B) + zb * p.strideB;
  const size_t coff = zb * p.strideC;
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / BK32;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform copy: scalar LDS destinations
  const TileOfs<TBM, NW> oA = tile_ofs<TBM, NW>(p.a_panels ? BK32 : p.lda, m0, p.M - 1, wave, lane);
  const TileOfs<BN, NW> oB = tile_ofs<BN, NW>(packed ? BK32 : p.ldb, n0, p.N - 1, wave, lane);
  const int kmulB = packed ? p.N * BK32 : BK32;
  const int kmulA = p.a_panels ? p.a_panels * BK32 : BK32;   // K-panel-major A: a_panels rows per panel
  stage_tile32_pre<TBM, NW>(A, 0, oA, smem, uwave);
  stage_tile32_pre<BN, NW>(B, 0, oB, smem + A_BYTES, uwave);
  int cur = 0;
  if constexpr (!TWOB) {
    // The K-extension ([T | Vs], Rp / 32 steps) is the tail of the SAME pipelined loop: its 64-byte rows are staged by LDS-DMA a
    // step ahead like every other step (it used to be a separate phase of plain loads behind the loop: an exposed memory
    // latency and two more barriers per tile).  Its row offsets are computed when it is staged, not kept through the loop.
    const int ntot = nk + (p.Rp >> 5);
    for (int kt = 0; kt < ntot; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      char* sA = smem + cur * SLOT;
      char* nA = smem + (cur ^ 1) * SLOT;
      if (kt + 1 < nk) {
        stage_tile32_pre<TBM, NW>(A, (kt + 1) * kmulA, oA, nA, uwave);
        stage_tile32_pre<BN, NW>(B, (kt + 1) * kmulB, oB, nA + A_BYTES, uwave);
      } else if (kt + 1 < ntot) {
        const int e = kt + 1 - nk;
        const TileOfs<TBM, NW> eA = tile_ofs<TBM, NW>(p.Rp, m0, p.M - 1, wave, lane);
        const TileOfs<BN, NW> eB = tile_ofs<BN, NW>(p.Rp, n0, p.N - 1, wave, lane);
        stage_tile32_pre<TBM, NW>(static_cast<const bf16*>(p.A2), e * 32, eA, nA, uwave);
        stage_tile32_pre<BN, NW>(static_cast<const bf16*>(p.B2), e * 32, eB, nA + A_BYTES, uwave);
      }
      mma_tile32<MI>(sA, sA + A_BYTES, acc, wr, wc, lane);
      cur ^= 1;
    }
  } else {
    // B3 has B's shape and ldb: the same row offsets, only the base pointer (a scalar) differs
    const bf16* __restrict__ B3 = static_cast<const bf16*>(p.B3);
    stage_tile32_pre<BN, NW>(B3, 0, oB, smem + A_BYTES + B32_BYTES, uwave);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      char* sA = smem + cur * SLOT;
      if (kt + 1 < nk) {
        char* nA = smem + (cur ^ 1) * SLOT;
        stage_tile32_pre<TBM, NW>(A, (kt + 1) * kmulA, oA, nA, uwave);
        stage_tile32_pre<BN, NW>(B, (kt + 1) * kmulB, oB, nA + A_BYTES, uwave);
        stage_tile32_pre<BN, NW>(B3, (kt + 1) * kmulB, oB, nA + A_BYTES + B32_BYTES, uwave);
      }
      mma_tile32_2b<MI>(sA, sA + A_BYTES, sA + A_BYTES + B32_BYTES, acc, wr, wc, lane);
      cur ^= 1;
    }
  }
  if constexpr (TWOB) {   // (the two-operand loop keeps the separate extension phase)
    for (int kk = 0; kk < (p.Rp >> 5); ++kk) {
      __syncthreads();
      stage_ext32<TBM, NW>(static_cast<const bf16*>(p.A2), p.Rp, m0, p.M - 1, kk, smem, tid);
      stage_ext32<BN, NW>(static_cast<const bf16*>(p.B2), p.Rp, n0, p.N - 1, kk, smem + A_BYTES, tid);
      __syncthreads();
      mma_tile32<MI>(smem, smem + A_BYTES, acc, wr, wc, lane);
    }
  }
  // epilogue.  bf16 outputs of interior wave tiles: the fast path of gemm_epilogue.h (values converted in the accumulator
  // layout, 2-byte LDS transposition); everything else: NPASS passes of HALF rows through a wave-private [HALF][64] fp32 image
  constexpr int NPASS = (MI % 2) ? MI : (((NW == 8 && MI == 4) || MI == 8) ? 4 : 2);
  constexpr int HALF = MI * 16 / NPASS;
  // one LDS region per wave for BOTH epilogue paths (in an edge tile some waves take the fast path and others the generic one)
  constexpr int WAVE_STG = HALF * 64 * 4 > EPI_FAST_WAVE_BYTES ? HALF * 64 * 4 : EPI_FAST_WAVE_BYTES;
  __syncthreads();
  STAMP(1);
  if constexpr (ER) {
    static_assert(EPI == CARA_EPI_MULH && NW == 4 && !TWOB, "epilogue riders: the fc2 dX product");
    // LDS of the riders' epilogue (the launch asks for ER_LDS_BYTES if that is more than the K loop's slots): per wave a 16 x 64 fp32
    // staging image + 256 B for the column sums, and the two bf16 images
    constexpr int ESTG = ER_STG_BYTES;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 rv[4], ru[4];
    float cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rv[j] = ru[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      cs[j] = 0.f;
    }
    float* stg = reinterpret_cast<float*>(smem + wave * ESTG);
    char* img = smem + 4 * ESTG + wave * ER_WAVE_BYTES;
    const int mw = m0 + wr * (MI * 16);
    if (mw + MI * 16 <= p.M) epilogue_mulh_riders<MI, true>(p, acc, stg, img, mw, n0 + wc * 64, lane, rv, ru, cs);   // (wave-uniform)
    else epilogue_mulh_riders<MI, false>(p, acc, stg, img, mw, n0 + wc * 64, lane, rv, ru, cs);
    // column sums: add the four row groups of the wave (lane bits 4, 5); lanes 0 .. 15 then hold columns 16 j + lane
    if (p.er_colsum) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cs[j] += __shfl_xor(cs[j], 16, 64);
        cs[j] += __shfl_xor(cs[j], 32, 64);
      }
    }
    // the two wave rows of a column: row 1 leaves its sums in its own staging area and image area, row 0 adds them, lays the
    // sums out as the slab (64 columns x 16 floats) in ITS areas and stores whole 1-KiB runs, 16 bytes per lane
    f32x4* xv = reinterpret_cast<f32x4*>(stg);
    f32x4* xu = reinterpret_cast<f32x4*>(img);
    float* xc = reinterpret_cast<float*>(smem + wave * ESTG + 4096);
    if (wr == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[j * 64 + lane] = rv[j];
        xu[j * 64 + lane] = ru[j];
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xc[j * 16 + lane] = cs[j];
      }
    }
    __syncthreads();
#if defined(CARA_ER_ABLATE) && (CARA_ER_ABLATE & 2)   // timing diagnostic: without the slab stores
    if (wr == 0 && p.M < 0) {
#else
    if (wr == 0) {
#endif
      const f32x4* pv = reinterpret_cast<const f32x4*>(smem + (wave + 2) * ESTG);
      const f32x4* pu = reinterpret_cast<const f32x4*>(smem + 4 * ESTG + (wave + 2) * ER_WAVE_BYTES);
      const float* pc = reinterpret_cast<const float*>(smem + (wave + 2) * ESTG + 4096);
      const int colblocks = p.N >> 6, tiles_m = nwg / tiles_n;
      const size_t blk = (size_t)tm * colblocks + ((n0 >> 6) + wc);
      float* sv = static_cast<float*>(p.er_slabs_v) + blk * (64 * 16);
      float* su = static_cast<float*>(p.er_slabs_u) + blk * (64 * 16);
      float* lv = reinterpret_cast<float*>(stg);
      float* lu = reinterpret_cast<float*>(img);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a = pv[j * 64 + lane], b = pu[j * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lv[(j * 16 + fq * 4 + r) * 16 + fr] = rv[j][r] + a[r];
          lu[(j * 16 + fq * 4 + r) * 16 + fr] = ru[j][r] + b[r];
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(lv + (t * 64 + lane) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(lu + (t * 64 + lane) * 4);
        *reinterpret_cast<f32x4*>(sv + (t * 64 + lane) * 4) = a;
        *reinterpret_cast<f32x4*>(su + (t * 64 + lane) * 4) = b;
      }
      if (p.er_colsum && lane < 16) {
        float* cv = static_cast<float*>(p.er_slabs_v) + (size_t)tiles_m * colblocks * (64 * 32) + blk * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j * 16 + lane] = cs[j] + pc[j * 16 + lane];
      }
    }
    STAMP_END();
    return;
  }
  if constexpr ((EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU) && (MI == 4 || MI == 8) && NW == 4) {
    const int mw = m0 + wr * (MI * 16), nw = n0 + wc * 64;
    if (mw + MI * 16 <= p.M && nw + 64 <= p.N && (p.ldc & 7) == 0) {   // wave-uniform
#pragma unroll
      for (int q = 0; q < MI / 4; ++q)
        epilogue_fast_bf16<EPI>(p, *reinterpret_cast<const f32x4(*)[4][4]>(&acc[q * 4]), smem + wave * WAVE_STG, mw + q * 64, nw, lane, coff);
      STAMP_END();
      return;
    }
  } else if constexpr (EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU || EPI == CARA_EPI_GELU_DG) {
    const int mw = m0 + wr * (MI * 16), nw = n0 + wc * 64;
    if (mw + MI * 16 <= p.M && nw + 64 <= p.N && (p.ldc & 7) == 0) {   // wave-uniform
      epilogue_fast_bf16_rt<EPI, MI>(p, acc, smem + wave * WAVE_STG, mw, nw, lane, coff);
      STAMP_END();
      return;
    }
  }
  float* stg = reinterpret_cast<float*>(smem + wave * WAVE_STG);
  if constexpr (EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU || EPI == CARA_EPI_MULH) {
    // interior wave tiles: the epilogue's input operand requested a pass group ahead (gemm_epilogue.h)
    const int mw = m0 + wr * (MI * 16), nw = n0 + wc * 64;
    const bool ok = mw + MI * 16 <= p.M && nw + 64 <= p.N && (p.ldc & 7) == 0 && coff == 0 && (EPI != CARA_EPI_MULH || !p.bias) &&
                    (EPI != CARA_EPI_RESID || !p.rowscale || p.rows_per_sample >= MI * 16) && (!p.bias || (nw & 3) == 0);
    if (ok) {   // wave-uniform
      epilogue_interior_aux<EPI, MI, 1>(p, acc, stg, mw, nw, lane);
      STAMP_END();
      return;
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int half = 0; half < NPASS; ++half) {
#pragma unroll
    for (int i = 0; i < MI / NPASS; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[half * (MI / NPASS) + i][j][r];
    asm volatile("" ::: "memory");   // (scalar stores, 16-byte loads of the same image: keep the compiler from reordering them)
    epilogue_rows<EPI, HALF>(p, stg, m0 + wr * (MI * 16) + half * HALF, n0 + wc * 64, lane, coff);
    asm volatile("" ::: "memory");
  }
  STAMP_END();
}

// MI = 5: 160 x 128 tiles (80 x 64 per wave) for the products with N >= 3072 -- their 99 x 24 = 2376 tiles of 128 rows are 2.32
// rounds of the 1024 workgroup slots (the last third of the launch runs at a fraction of the occupancy, tools/gemm_stamps.py),
// 79 x 24 = 1896 tiles of 160 rows are 1.85, and a tile stages 10 % fewer bytes per flop; 36 KiB of LDS, still four per CU
// (the 160-row tile with an epilogue that reads a second operand and no riding products -- not a product of the model -- would spill
// a few registers at four workgroups per CU: it gets three)
template <int EPI, bool TWOB = false, int MI = 4, bool ER = false>
__global__ __launch_bounds__(256, (TWOB || ER || (MI == 5 && (EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU || EPI == CARA_EPI_MULH))) ? 3 : 4) void gemm32_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm32_body<EPI, MI, 4, TWOB, ER>(p, tiles_n, nwg, gm, blockIdx.x, blockIdx.y, smem);
}

// 8-wave workgroups (4 along M x 2 along N, 64-column wave tiles): MI = 3 -> 192 x 128 tiles, MI = 4 -> 256 x 128.  For the
// N = 768 products: their 594 tiles of 128 rows put THREE workgroups on 82 of the 256 CUs and two on the rest, and a CU's
// K loops run at what its load path delivers, so the launch lasts as long as three tiles on one CU;
// 66 x 6 = 396 tiles of 192 rows are at most two per CU, 0.83 of the staged bytes per flop, and two 8-wave workgroups keep
// 16 waves feeding the path.
template <int EPI, int MI>
__global__ __launch_bounds__(512, 4) void gemm32w8_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm32_body<EPI, MI, 8>(p, tiles_n, nwg, gm, blockIdx.x, 0, smem);
}

// The dX GEMM of a linear and the two transposed skinny products of the SAME linear (dU = X^T G', dVs = dY^T T) in
// one grid: blocks [0, nts) are tskinny blocks (HBM-bound, one LDS stage per wave), the rest GEMM tiles (MFMA-bound).
// The products used to run on a side stream under the GEMM, which costs a fork (an event record = 3..7 us of idle
// chip, 44 of them per backward pass) and overlaps only as well as two queues happen to interleave; as one launch
// there is no event at all and the dispatcher mixes the two kinds of workgroup on every CU.  Needs Rp = 32 products
// (84 VGPRs; the Rp = 64 form needs 136) and 36 KiB of LDS per workgroup (still four per CU).
// NT = Rp / 16 of the riding products: 2 (rank <= 32: four workgroups per CU) or 4 (rank <= 64: the products' 16 more
// accumulator tiles take the kernel to 136 VGPRs, three workgroups per CU -- still far better than the products as a
// launch of their own behind the GEMM, 55 us per pair at rank 64)
template <int EPI, bool COLSUM, int MI = 4, int NT = 2, bool ER = false>
#ifndef CARA_TS_WG   // (timing diagnostic: workgroups per CU of the riders-carrying kernels that default to four)
#define CARA_TS_WG 4
#endif
#ifndef CARA_ER_WG
#define CARA_ER_WG 3
#endif
__global__ __launch_bounds__(256, ER ? CARA_ER_WG : (NT <= 2 ? CARA_TS_WG : 3)) void gemm32_ts_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int gm,
                                                           const TsProblem t0, const TsProblem t1, const int ldg, const int Mts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the products' blocks sit BEHIND the GEMM tiles: they fill the slots the GEMM's last, partly filled round leaves
  // free (in front of the tiles: +0.05 ms per step; spread among them: +0.7 ms)
  const int b = blockIdx.x;
  if (b >= nwg) {
    STAMP(0);
    tskinny_body<NT, COLSUM, 1>(t0, t1, ldg, Mts, b - nwg, smem);
    STAMP(1);
    STAMP_END();
  } else {
    gemm32_body<EPI, MI, 4, false, ER>(p, tiles_n, nwg, gm, b, 0, smem);
  }
}

// ---------------------------------------------------------------------------------------------
// The same 128x128x32 kernel with the WHOLE adapter inside (cara_gemm_args::Ut): besides its 128 x 128 outputs
// every workgroup accumulates T[128 rows, 32] = A_rows . Ut^T on the operand tiles it streams anyway (a 2-KiB
// slab of Ut per K step: +12 % staged bytes, +25 % MFMAs in a loop that waits on staging), rounds T to bf16 into
// the LDS image the K-extension step reads, and the tiles of column 0 write T / Tt out for the backward.  The
// separate cara_skinny_xu pass (a full re-read of A, a launch, a dependency) is gone.
// ---------------------------------------------------------------------------------------------
// NU = Rp / 32: 1 (rank <= 32) or 2 (rank <= 64: a 4-KiB slab of Ut per K step, eight T tiles per wave, two extension steps; 140
// VGPRs, three workgroups per CU -- the kernel serves the N = 768 products, whose 594 tiles never put more than three on a CU)
// HALFT (NU = 1 only; cara_gemm_args::Ut_rank <= 16): columns 16 .. 31 of T are zero by construction (rows >= rank of Ut are zero),
// so a workgroup stages 16 rows of Ut per K step instead of 32 and every wave computes TWO T tiles (rows of its wave row:
// tiles 2 wc, 2 wc + 1; columns 0 .. 15) instead of four: 18 MFMAs per K step instead of 20, one 1-KiB piece less.  All four
// waves keep the same instruction stream (skipping the second wave column's T tiles behind a branch cost 57 %).
template <int EPI, int NU = 1, bool HALFT = false>
__device__ __forceinline__ void gemm32ft_body(const cara_gemm_args& p, const int tiles_n, const int nwg, const int gm, const int block, char* smem) {
  static_assert(!HALFT || NU == 1, "the 16-column form exists for Rp = 32 only");
  constexpr int TBM = 128;
  constexpr int NTT = HALFT ? 2 : 4;   // T tiles per wave
  constexpr int A_BYTES = TBM * BK32 * 2, U_BYTES = NU * 32 * BK32 * 2;
  constexpr int SLOT = A_BYTES + B32_BYTES + U_BYTES;   // 18 KiB; two slots = 36 KiB, still 4 workgroups per CU (NU = 2: 20 / 40 KiB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = gm < 0 ? xcd_remap_cu(block, nwg) : xcd_remap(block, nwg);   // (gm < 0: CARA_GEMM_CUSHARE, plain order)
  int tm, tn;
  if (gm <= 1) {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  } else {
    const int tiles_m = nwg / tiles_n;
    const int per_group = gm * tiles_n;
    const int gid = tile / per_group, rem = tile - gid * per_group;
    const int first = gid * gm;
    const int gsz = (tiles_m - first) < gm ? (tiles_m - first) : gm;
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  }
  const int m0 = tm * TBM, n0 = tn * BN;
  const bf16* __restrict__ A = static_cast<const bf16*>(p.A);
  const bool packed = p.Bp != nullptr;   // K-panel-major image of B (cara_gemm_args::Bp)
  const bf16* __restrict__ B = static_cast<const bf16*>(packed ? p.Bp : p.B);
  const int kmulB = packed ? p.N * BK32 : BK32;
  const bf16* __restrict__ Ut = static_cast<const bf16*>(p.Ut);
  f32x4 acc[4][4], accg[NTT][NU];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NTT; ++i)
#pragma unroll
    for (int c = 0; c < NU; ++c) accg[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the Rp x 32 slab of Ut of one K step = 2 NU one-KiB pieces (16 rows x 64 B), issued by waves 0 .. 2 NU - 1
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const unsigned offU = (unsigned)((wave & (2 * NU - 1)) * 16 + (lane >> 2)) * (unsigned)(p.K * 2) +
                        (unsigned)((((lane & 3) ^ ((((lane >> 2) >> 3) & 1) * 3))) * 16);
  auto stage_u = [&](int k0, char* dst) {
    if (uwave < (HALFT ? 1 : 2 * NU)) glds16(reinterpret_cast<const char*>(Ut + k0) + offU, dst + uwave * 1024);
  };
  const int nk = p.K / BK32;
  const TileOfs<TBM, 4> oA = tile_ofs<TBM, 4>(p.a_panels ? BK32 : p.lda, m0, p.M - 1, wave, lane);
  const int kmulA = p.a_panels ? p.a_panels * BK32 : BK32;
  const TileOfs<BN, 4> oB = tile_ofs<BN, 4>(packed ? BK32 : p.ldb, n0, p.N - 1, wave, lane);
  stage_tile32_pre<TBM, 4>(A, 0, oA, smem, uwave);
  stage_tile32_pre<BN, 4>(B, 0, oB, smem + A_BYTES, uwave);
  stage_u(0, smem + A_BYTES + B32_BYTES);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* sA = smem + cur * SLOT;
    char* nA = smem + (cur ^ 1) * SLOT;
    if (kt + 1 < nk) {
      stage_tile32_pre<TBM, 4>(A, (kt + 1) * kmulA, oA, nA, uwave);
      stage_tile32_pre<BN, 4>(B, (kt + 1) * kmulB, oB, nA + A_BYTES, uwave);
      stage_u((kt + 1) * BK32, nA + A_BYTES + B32_BYTES);
    } else {
      // last K step: the extension's B operand (Vs rows of this column tile, 64-byte rows) goes to the free slot by LDS-DMA
      // under this step's MFMAs (it used to be plain loads behind the loop: an exposed memory latency per tile)
      const TileOfs<BN, 4> eB = tile_ofs<BN, 4>(32 * NU, n0, p.N - 1, wave, lane);
      stage_tile32_pre<BN, 4>(static_cast<const bf16*>(p.B2), 0, eB, nA + A_BYTES, uwave);
    }
    const char* sB = sA + A_BYTES;
    const char* sU = sB + B32_BYTES;
    bf16x8 a[4], b[4], bu[NU];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sA + swz32(wr * 64 + i * 16 + fr, fq));
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sB + swz32(wc * 64 + j * 16 + fr, fq));
#pragma unroll
    for (int c = 0; c < NU; ++c) bu[c] = *reinterpret_cast<const bf16x8*>(sU + swz32(((HALFT ? 0 : wc) * NU + c) * 16 + fr, fq));   // this wave's 16 NU columns of T
    if constexpr (HALFT) {
      // T tiles (2 wc, 2 wc + 1) x column tile 0: the A fragments picked by a wave-uniform select, no branch
      const bf16x8 ah0 = wc ? a[2] : a[0], ah1 = wc ? a[3] : a[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (i == 1) accg[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bu[0], accg[0][0], 0, 0, 0);
        if (i == 3) accg[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bu[0], accg[1][0], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NU; ++c) accg[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bu[c], accg[i][c], 0, 0, 0);
      }
    }
    cur ^= 1;
  }
  // T tile (rows wr*64 .., columns wc*16 NU ..) -> bf16 -> the A image(s) of the extension step(s): image 0 in the slot the
  // loop left free, next to the Vs rows already on their way there; NU = 2: image 1 (T columns 32..63, all of them wave
  // column 1's) in the other slot, whose last K step every wave has read past the barrier below, next to the second
  // 32 columns of the Vs rows, requested here.  Column-0 tiles also write T (and its transpose) to global for the
  // backward's transposed skinny products
  __syncthreads();
  char* sE = smem + cur * SLOT;
  char* sO = smem + (cur ^ 1) * SLOT;
  if constexpr (NU == 2) {
    const TileOfs<BN, 4> eB = tile_ofs<BN, 4>(32 * NU, n0, p.N - 1, wave, lane);
    stage_tile32_pre<BN, 4>(static_cast<const bf16*>(p.B2), 32, eB, sO + A_BYTES, uwave);
  }
  {
    bf16* T = static_cast<bf16*>(p.T_out);
    bf16* Tt = static_cast<bf16*>(p.Tt_out);
    char* img = (NU == 2 && wc == 1) ? sO : sE;   // the image that holds this wave's T columns
#pragma unroll
    for (int c = 0; c < (HALFT ? 2 : NU); ++c) {            // (HALFT: c = 0 the computed columns 0 .. 15, c = 1 the zero columns 16 .. 31)
      const int icol = HALFT ? c * 16 + fr : (NU == 2 ? c : wc) * 16 + fr;      // column inside its 32-column image
      const int col = NU == 2 ? wc * 32 + icol : icol;     // column of T
#pragma unroll
      for (int i = 0; i < NTT; ++i) {
        const int row0 = wr * 64 + (HALFT ? wc * 2 + i : i) * 16 + fq * 4;
        const f32x4 av = (HALFT && c == 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : accg[i][HALFT ? 0 : c];
        bf16x4 tv = {(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<bf16*>(img + swz32(row0 + r, icol >> 3) + (icol & 7) * 2) = tv[r];
        if (tn == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (m0 + row0 + r < p.M) T[(size_t)(m0 + row0 + r) * (32 * NU) + col] = tv[r];
          if (Tt) {
            if (m0 + row0 + 4 <= p.M) {
              *reinterpret_cast<bf16x4*>(Tt + (size_t)col * p.ldt + m0 + row0) = tv;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (m0 + row0 + r < p.M) Tt[(size_t)col * p.ldt + m0 + row0 + r] = tv[r];
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the Vs rows have landed
  __syncthreads();
  mma_tile32<4>(sE, sE + A_BYTES, acc, wr, wc, lane);
  if constexpr (NU == 2) mma_tile32<4>(sO, sO + A_BYTES, acc, wr, wc, lane);
  // epilogue: two 32-row halves through a wave-private [32][64] fp32 image
  constexpr int HALF = 32;
  __syncthreads();
  float* stg = reinterpret_cast<float*>(smem) + wave * (HALF * 64);
  if constexpr (EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU) {
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const bool ok = mw + 64 <= p.M && nw + 64 <= p.N && (p.ldc & 7) == 0 &&
                    (EPI == CARA_EPI_DGELU || !p.rowscale || p.rows_per_sample >= 64) && (!p.bias || (nw & 3) == 0);
    if (ok) {   // wave-uniform
      epilogue_interior_aux<EPI, 4, 2>(p, acc, stg, mw, nw, lane);
      return;
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[half * 2 + i][j][r];
    asm volatile("" ::: "memory");
    epilogue_rows<EPI, HALF>(p, stg, m0 + wr * 64 + half * HALF, n0 + wc * 64, lane);
    asm volatile("" ::: "memory");
  }
}

template <int EPI, int NU = 1, bool HALFT = false>
__global__ __launch_bounds__(256, NU == 1 ? 4 : 3) void gemm32ft_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm32ft_body<EPI, NU, HALFT>(p, tiles_n, nwg, gm, blockIdx.x, smem);
}

// the adapter-inside GEMM carrying a pair of transposed skinny products (of ANOTHER linear: this launch only now produces
// the G' its own products would read) behind its tiles, as gemm32_ts_kernel does
// (NTS: column tiles of 16 the riding products compute -- 2 NU, or 1 at rank <= 16; HALFT: the 16-column form of the adapter inside)
template <int EPI, int NU, bool COLSUM, int NTS = 2 * NU, bool HALFT = false>
__global__ __launch_bounds__(256, NU == 1 ? 4 : 3) void gemm32ft_ts_kernel(const cara_gemm_args p, const int tiles_n, const int nwg, const int gm,
                                                                           const TsProblem t0, const TsProblem t1, const int ldg, const int Mts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (b >= nwg) tskinny_body<NTS, COLSUM, 1>(t0, t1, ldg, Mts, b - nwg, smem);
  else gemm32ft_body<EPI, NU, HALFT>(p, tiles_n, nwg, gm, b, smem);
}

static int group_m(int tiles_n);
struct TsPair;
template <int EPI>
int launch32ft(const cara_gemm_args* a, hipStream_t st, const TsPair* ts = nullptr);

// rows per supertile (1 = plain row-major order)
static int group_m(int tiles_n) {
  // measured with rocprofv3 FETCH_SIZE (x2 gfx950 correction), per launch, plain order -> groups of 8:
  // fc1 fwd (24 column tiles) 224 -> 133 MB, fc2 bwd 297 -> 207 MB, but the 6-column products
  // 82 -> 113 MB and qkv (18 columns) flat: group only when there are many column tiles
  static const int gm_wide = [] { const char* e = getenv("CARA_GEMM_GM"); return e ? atoi(e) : 8; }();       // (A/B: rows per supertile of the wide products)
  static const int gm_minn = [] { const char* e = getenv("CARA_GEMM_GM_MINT"); return e ? atoi(e) : 20; }();   // (A/B: fewest column tiles that get supertiles)
  if (tiles_n >= gm_minn) return gm_wide > 0 ? gm_wide : 1;
  // CARA_GEMM_CUSHARE=1: the narrow products (plain order) with the workgroups of a CU on consecutive tiles (xcd_remap_cu)
  static const int cushare = [] { const char* e = getenv("CARA_GEMM_CUSHARE"); return e ? atoi(e) : 0; }();
  return (cushare && tiles_n <= 8) ? -1 : 1;
}

// the transposed skinny products a GEMM launch can carry (cara_gemm_with_tskinny)
struct TsPair {
  TsProblem a, b;
  int ldg, M;
  bool any_cs;
  int nt;   // column tiles of 16 the products compute: Rp / 16 = 2 or 4, or 1 at Rp = 32 and rank <= 16
};

// one launch of the GEMM-with-riders kernel: COLSUM and NT (the products' Rp / 16) picked at run time
template <int EPI, int MI, bool ER = false>
void launch_ts(const cara_gemm_args* a, hipStream_t st, const TsPair* ts, int tiles_n, int nwg, int gm, int gemm_lds) {
  const int nts = ts->a.nblk + ts->b.nblk;
  const dim3 grid(nwg + nts), block(256);
#define TS_GO(CS, NT)                                                                                                       \
  do {                                                                                                                      \
    constexpr int RB = TsRing<NT, 1>::BLOCK_BYTES;                                                                          \
    const int lds = RB > gemm_lds ? RB : gemm_lds;                                                                          \
    hipLaunchKernelGGL((gemm32_ts_kernel<EPI, CS, MI, NT, ER>), grid, block, lds, st, *a, tiles_n, nwg, gm, ts->a, ts->b, ts->ldg, ts->M); \
  } while (0)
  if constexpr (ER) {   // (epilogue riders: rank <= 16, so are the products that ride as workgroups)
    if (ts->any_cs) TS_GO(true, 1); else TS_GO(false, 1);
    return;
  }
  if (ts->nt == 4) {
    if (ts->any_cs) TS_GO(true, 4); else TS_GO(false, 4);
  } else if (ts->nt == 1) {   // rank <= 16: the products compute 16 of their 32 columns (16-wide slabs)
    if (ts->any_cs) TS_GO(true, 1); else TS_GO(false, 1);
  } else {
    if (ts->any_cs) TS_GO(true, 2); else TS_GO(false, 2);
  }
#undef TS_GO
}

template <int EPI>
int launch32ft(const cara_gemm_args* a, hipStream_t st, const TsPair* ts) {
  const int tiles_n = (a->N + BN - 1) / BN;
  const int gm = group_m(tiles_n);
  const int nwg = ((a->M + 127) / 128) * tiles_n;
  // consumers read Tt in whole 32-row steps: keep columns [M, roundup32(M)) zero, as cara_skinny_xu does
  const int m32 = (a->M + 31) / 32 * 32;
  if (a->Tt_out && m32 > a->M && m32 <= a->ldt &&
      hipMemset2DAsync(static_cast<bf16*>(a->Tt_out) + a->M, (size_t)a->ldt * 2, 0, (size_t)(m32 - a->M) * 2, a->Rp, st) != hipSuccess)
    return CARA_E_LAUNCH;
  const int lds1 = 2 * (128 * BK32 * 2 + B32_BYTES + 32 * BK32 * 2), lds2 = 2 * (128 * BK32 * 2 + B32_BYTES + 64 * BK32 * 2);
  if (ts) {   // (plain bf16 output only: the backward's fc1 / qkv dX)
    if constexpr (EPI == CARA_EPI_BF16) {
      const bool half = a->Rp == 32 && ts->nt == 1;   // rank <= 16: one r-tile in the riding products, 16 columns of the adapter inside
      if (ts->nt != a->Rp / 16 && !half) return CARA_E_ARG;
      if (half && !(a->Ut_rank > 0 && a->Ut_rank <= 16)) return CARA_E_ARG;
      const dim3 grid(nwg + ts->a.nblk + ts->b.nblk), block(256);
      constexpr int RB = TsRing<2, 1>::BLOCK_BYTES;   // the same for every NT (two-pass combine)
#define FT_GO(NU, CS, L, NTS, HT)                                                                                           \
      hipLaunchKernelGGL((gemm32ft_ts_kernel<EPI, NU, CS, NTS, HT>), grid, block, (RB > L ? RB : L), st, *a, tiles_n, nwg, gm, ts->a, ts->b, ts->ldg, ts->M)
      if (a->Rp == 64) {
        if (ts->any_cs) FT_GO(2, true, lds2, 4, false); else FT_GO(2, false, lds2, 4, false);
      } else if (half) {
        if (ts->any_cs) FT_GO(1, true, lds1, 1, true); else FT_GO(1, false, lds1, 1, true);
      } else {
        if (ts->any_cs) FT_GO(1, true, lds1, 2, false); else FT_GO(1, false, lds1, 2, false);
      }
#undef FT_GO
      CARA_CHECK_LAUNCH();
      return CARA_OK;
    } else {
      return CARA_E_ARG;
    }
  }
  if (a->Rp == 64) hipLaunchKernelGGL((gemm32ft_kernel<EPI, 2>), dim3(nwg), dim3(256), lds2, st, *a, tiles_n, nwg, gm);
  else if (a->Ut_rank > 0 && a->Ut_rank <= 16) hipLaunchKernelGGL((gemm32ft_kernel<EPI, 1, true>), dim3(nwg), dim3(256), lds1, st, *a, tiles_n, nwg, gm);
  else hipLaunchKernelGGL((gemm32ft_kernel<EPI>), dim3(nwg), dim3(256), lds1, st, *a, tiles_n, nwg, gm);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// rows of the tile that carries epilogue riders: CARA_ER_ROWS = 128 (default: 64 accumulator registers leave room for the riders' 32 +
// the operand buffers inside the 168 of three workgroups per CU) or 160
static int er_rows() {
  static const int v = [] { const char* e = getenv("CARA_ER_ROWS"); return e ? atoi(e) : 128; }();
  return v == 160 ? 160 : 128;
}
// the 160-row tile takes the widest products (launch32)
static bool tile160(const cara_gemm_args* a) {
  static const int bm = [] { const char* e = getenv("CARA_GEMM_BM"); return e ? atoi(e) : 160; }();
  static const int bm_minn = [] { const char* e = getenv("CARA_GEMM_BM_MINN"); return e ? atoi(e) : 3072; }();   // A/B: 2304 adds qkv forward
  return bm == 160 && a->N >= bm_minn && a->M > 1024 && a->batch <= 1 && !a->B3;
}

template <int EPI>
int launch32(const cara_gemm_args* a, hipStream_t st, const TsPair* ts = nullptr) {
  const int tiles_n = (a->N + BN - 1) / BN;
  const int gm = group_m(tiles_n);
  const int nwg = ((a->M + 127) / 128) * tiles_n;
  constexpr int GEMM_LDS = 2 * (128 * BK32 * 2 + B32_BYTES);
  // CARA_GEMM_BM=160: the 160-row tile for the widest products (A/B)
  // The 160-row tile for the widest products (N >= 3072: fc1 forward, fc2 dX).  CARA_GEMM_BM=128 keeps the 128-row tile (A/B runs:
  // 9.22 -> 9.08 and 9.37 -> 9.28 ms per step on two boxes; for the N = 768 products, whose 594 / 474 tiles are a single
  // round either way, it made no difference in the step and stays off)
  // launches that carry riding products: the dX GEMMs' epilogues only
  constexpr bool TS_EPI = EPI == CARA_EPI_BF16 || EPI == CARA_EPI_DGELU || EPI == CARA_EPI_MULH;
  if (tile160(a)) {
    constexpr int LDS160 = 2 * (160 * BK32 * 2 + B32_BYTES);
    const int nwg5 = ((a->M + 159) / 160) * tiles_n;
    if constexpr (EPI == CARA_EPI_MULH) {
      if (a->er_Tt) {   // epilogue riders (checked by the caller: cara_gemm_epi_rider_chunks)
        if (er_rows() == 160) {
          constexpr int L = LDS160 > ER_LDS_BYTES ? LDS160 : ER_LDS_BYTES;
          if (ts) launch_ts<EPI, 5, true>(a, st, ts, tiles_n, nwg5, gm, L);
          else hipLaunchKernelGGL((gemm32_kernel<EPI, false, 5, true>), dim3(nwg5), dim3(256), L, st, *a, tiles_n, nwg5, gm);
        } else {
          constexpr int L = GEMM_LDS > ER_LDS_BYTES ? GEMM_LDS : ER_LDS_BYTES;
          if (ts) launch_ts<EPI, 4, true>(a, st, ts, tiles_n, nwg, gm, L);
          else hipLaunchKernelGGL((gemm32_kernel<EPI, false, 4, true>), dim3(nwg), dim3(256), L, st, *a, tiles_n, nwg, gm);
        }
        CARA_CHECK_LAUNCH();
        return CARA_OK;
      }
    }
    if (ts) {
      if constexpr (TS_EPI) launch_ts<EPI, 5>(a, st, ts, tiles_n, nwg5, gm, LDS160);
      else return CARA_E_ARG;
    } else {
      hipLaunchKernelGGL((gemm32_kernel<EPI, false, 5>), dim3(nwg5), dim3(256), LDS160, st, *a, tiles_n, nwg5, gm);
    }
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  if (ts) {
    if constexpr (TS_EPI) launch_ts<EPI, 4>(a, st, ts, tiles_n, nwg, gm, GEMM_LDS);
    else return CARA_E_ARG;
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  // CARA_GEMM_W8 = 3 / 4: 192 x 128 / 256 x 128 tiles of 8 waves for the narrow products (N <= 1024)
  static const int w8 = [] { const char* e = getenv("CARA_GEMM_W8"); return e ? atoi(e) : 0; }();
  if ((w8 == 3 || w8 == 4) && a->N <= 1024 && a->M > 1024 && a->batch <= 1 && !a->B3) {
    if (w8 == 3) {
      const int nw = ((a->M + 191) / 192) * tiles_n;
      hipLaunchKernelGGL((gemm32w8_kernel<EPI, 3>), dim3(nw), dim3(512), 2 * (192 * BK32 * 2 + B32_BYTES), st, *a, tiles_n, nw, gm);
    } else {
      const int nw = ((a->M + 255) / 256) * tiles_n;
      hipLaunchKernelGGL((gemm32w8_kernel<EPI, 4>), dim3(nw), dim3(512), 2 * (256 * BK32 * 2 + B32_BYTES), st, *a, tiles_n, nw, gm);
    }
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  const int nb = a->batch > 1 ? a->batch : 1;
  if (a->B3) hipLaunchKernelGGL((gemm32_kernel<EPI, true>), dim3(nwg, nb), dim3(256), GEMM_LDS + 2 * B32_BYTES, st, *a, tiles_n, nwg, gm);
  else hipLaunchKernelGGL((gemm32_kernel<EPI>), dim3(nwg, nb), dim3(256), GEMM_LDS, st, *a, tiles_n, nwg, gm);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Few-row products (the last block's cls-row-only proj / fc1 / fc2 and their dX: M = batch rows): one 128-row tile
// per 128 columns is 6-24 workgroups each walking all of K (24-96 K steps back to back: 20-70 us of latency).
// With caller scratch the K loop is cut into slabs that run as ONE batched launch of the default kernel (fp32
// partial products into scratch), and a small finishing kernel adds the slabs in fixed order, the rank-R term
// T Vs^T, the bias, and applies the epilogue.
// ---------------------------------------------------------------------------------------------
// the tail of a few-row product for outputs (m, n .. n + 3), v = their A B^T sums: + the rank-R term T Vs^T, + bias, the epilogue
template <int EPI>
__device__ __forceinline__ void small_m_finish4(const cara_gemm_args& p, const int m, const int n, float (&v)[4]) {
  if (p.Rp > 0) {   // the adapter term, straight from T [M,Rp] and Vs [N,Rp]
    const bf16* t = static_cast<const bf16*>(p.A2) + (size_t)m * p.Rp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (n + k >= p.N) continue;
      const bf16* vs = static_cast<const bf16*>(p.B2) + (size_t)(n + k) * p.Rp;
      float d = 0.f;
      for (int r = 0; r < p.Rp; r += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(t + r), b = *reinterpret_cast<const bf16x8*>(vs + r);
#pragma unroll
        for (int j = 0; j < 8; ++j) d += (float)a[j] * (float)b[j];
      }
      v[k] += d;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] += (p.bias && n + k < p.N) ? p.bias[n + k] : 0.f;
  const size_t o = (size_t)m * p.ldc + n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (n + k >= p.N) continue;
    if constexpr (EPI == CARA_EPI_F32) {
      static_cast<float*>(p.C)[o + k] = v[k];
    } else if constexpr (EPI == CARA_EPI_RESID) {
      const float rs = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.f;
      static_cast<float*>(p.C)[o + k] = static_cast<const float*>(p.aux)[o + k] + rs * v[k];
    } else if constexpr (EPI == CARA_EPI_BF16) {
      static_cast<bf16*>(p.C)[o + k] = (bf16)v[k];
    } else if constexpr (EPI == CARA_EPI_GELU) {
      if (p.C2) static_cast<bf16*>(p.C2)[o + k] = (bf16)v[k];
      static_cast<bf16*>(p.C)[o + k] = (bf16)gelu_erf(v[k]);
    } else if constexpr (EPI == CARA_EPI_GELU_DG) {
      float g, gp;
      gelu_erf_both(v[k], g, gp);
      if (p.C2) static_cast<h16*>(p.C2)[o + k] = (h16)gp;
      static_cast<bf16*>(p.C)[o + k] = (bf16)g;
    } else if constexpr (EPI == CARA_EPI_MULH) {
      static_cast<bf16*>(p.C)[o + k] = (bf16)(v[k] * (float)static_cast<const h16*>(p.aux)[o + k]);
    } else {
      static_cast<bf16*>(p.C)[o + k] = (bf16)(v[k] * gelu_erf_grad((float)static_cast<const bf16*>(p.aux)[o + k]));
    }
  }
}

// The few-row product in ONE launch (the default; CARA_SMALL_M_DIRECT=0: the batched split-K launch + the finishing kernel below,
// 8 + 9 us per product, seven of them per step).  A workgroup of 8 waves owns ONE 16 x 16 output tile; its waves split K (wave w: the
// 32-wide K steps w, w + 8, ...), every MFMA operand fragment is one 16-byte global load per lane straight into registers (rows of
// A / W are 64-byte runs: no LDS staging for a product whose operands are read once), six K steps of loads in flight; the eight
// partial tiles meet in LDS, are added in a fixed order, and 64 threads finish four outputs of a row each (small_m_finish4).
// One tile per workgroup, not a column of tiles: a workgroup that walks all of K reads all K columns of its A rows, and a CU takes
// ~70 GB/s from the L2 -- with the four row tiles of M = 64 in one workgroup (393 KB of A at K = 3072) the launch lasted 23 us.
// 192 (N = dim) .. 768 (N = 4 dim) workgroups at M = 64.
// NW waves split K: 8 for the long products, 4 where K <= 1024 (three K steps per wave would be all launch and no work)
template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void small_m_direct_kernel(const cara_gemm_args p) {
  __shared__ float red[NW][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int nb = n0 + fr < p.N ? n0 + fr : p.N - 1;
  const int mb = m0 + fr < p.M ? m0 + fr : p.M - 1;
  const bf16* pb = static_cast<const bf16*>(p.B) + (size_t)nb * p.ldb + fq * 8;
  const bf16* pa = static_cast<const bf16*>(p.A) + (size_t)mb * p.lda + fq * 8;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / BK32;
  constexpr int UN = 6;
  int kt = wave;
  for (; kt + NW * (UN - 1) < nk; kt += NW * UN) {
    bf16x8 a[UN], b[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k0 = (kt + NW * u) * BK32;
      a[u] = *reinterpret_cast<const bf16x8*>(pa + k0);
      b[u] = *reinterpret_cast<const bf16x8*>(pb + k0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  if (kt < nk) {   // the rest: up to UN - 1 steps, requested together (wave-uniform)
    bf16x8 a[UN - 1], b[UN - 1];
#pragma unroll
    for (int u = 0; u < UN - 1; ++u) {
      const int k = kt + NW * u;
      const int k0 = (k < nk ? k : nk - 1) * BK32;
      a[u] = *reinterpret_cast<const bf16x8*>(pa + k0);
      b[u] = *reinterpret_cast<const bf16x8*>(pb + k0);
    }
#pragma unroll
    for (int u = 0; u < UN - 1; ++u)
      if (kt + NW * u < nk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);   // (wave-uniform)
  }
  // accumulator layout: acc[r] = C[m0 + 4 fq + r][n0 + fr]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(fq * 4 + r) * 16 + fr] = acc[r];
  __syncthreads();
  if (tid < 64) {   // four consecutive columns of a row per thread
    const int row = tid >> 2, c4 = (tid & 3) * 4;
    const int m = m0 + row;
    if (m < p.M) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][row * 16 + c4 + k];
        v[k] = t;
      }
      small_m_finish4<EPI>(p, m, n0 + c4, v);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void small_m_finish_kernel(const cara_gemm_args p, const float* __restrict__ slabs, const int nslab) {
  const int n4 = (p.N + 3) / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.M * n4) return;
  const int m = idx / n4, n = (idx - m * n4) * 4;
  const size_t slab_stride = (size_t)p.M * p.N;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = n + 4 <= p.N && (p.N & 3) == 0;
  for (int s = 0; s < nslab; ++s) {
    const float* src = slabs + s * slab_stride + (size_t)m * p.N + n;
    if (full) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += t[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (n + k < p.N) v[k] += src[k];
    }
  }
  small_m_finish4<EPI>(p, m, n, v);
}

// K slabs of at least 128 columns, at most 16 of them; 0 = not worth it
static int small_m_slabs(const cara_gemm_args* a) {
  if (a->M > 128 || a->K < 512 || !a->scratch || a->Ut || a->batch > 1 || a->B3) return 0;
  int s = a->K / 128;
  if (s > 16) s = 16;
  while (s > 1 && (a->K % (s * 64)) != 0) --s;   // equal slabs, each a multiple of 64 columns
  if (s < 2 || (size_t)s * a->M * a->N * sizeof(float) > a->scratch_bytes) return 0;
  return s;
}

template <int EPI>
static int launch_small_m(const cara_gemm_args* a, int nslab, hipStream_t st) {
  static const int direct = [] { const char* e = getenv("CARA_SMALL_M_DIRECT"); return e ? atoi(e) : 1; }();
  if (direct) {   // (B: the row-major weights, also where a K-panel-major image exists)
    const dim3 grid((a->N + 15) / 16, (a->M + 15) / 16);
    if (a->K > 1024) hipLaunchKernelGGL((small_m_direct_kernel<EPI, 8>), grid, dim3(512), 0, st, *a);
    else hipLaunchKernelGGL((small_m_direct_kernel<EPI, 4>), grid, dim3(256), 0, st, *a);
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  cara_gemm_args d = {};
  d.A = a->A; d.lda = a->lda; d.B = a->B; d.ldb = a->ldb; d.M = a->M; d.N = a->N; d.K = a->K / nslab;
  d.epi = CARA_EPI_F32; d.C = a->scratch; d.ldc = a->N;
  d.batch = nslab; d.strideA = d.K; d.strideB = d.K; d.strideC = (long long)a->M * a->N;
  const int rc = launch32<CARA_EPI_F32>(&d, st);
  if (rc != CARA_OK) return rc;
  const int n4 = (a->N + 3) / 4;
  hipLaunchKernelGGL((small_m_finish_kernel<EPI>), dim3((a->M * n4 + 255) / 256), dim3(256), 0, st, *a, static_cast<const float*>(a->scratch), nslab);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// B [N, K] row-major -> K-panel-major [K/32][N][32] (cara_gemm_args::Bp): one 16-byte chunk per thread
__global__ __launch_bounds__(256) void pack_b_panels_kernel(const bf16* __restrict__ B, int ldb, int N, int K, bf16* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)N * (K / 8);
  if (i >= total) return;
  const size_t panel = i / ((size_t)N * 4), rem = i - panel * (size_t)N * 4;
  const size_t n = rem >> 2, c = rem & 3;
  *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(B + n * ldb + panel * 32 + c * 8);
}

extern "C" int cara_pack_b_panels(const void* B, int ldb, int N, int K, void* out, void* stream) {
  if (!B || !out || N <= 0 || K <= 0 || (K % 32) || ldb < K || (ldb & 7)) return CARA_E_ARG;
  const size_t total = (size_t)N * (K / 8);
  hipLaunchKernelGGL(pack_b_panels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const bf16*>(B), ldb, N, K, static_cast<bf16*>(out));
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// ---------------------------------------------------------------------------------------------
// C[z][m, n] (fp32) = sum over the rows k of slab z of At[k, m] * Bt[k, n]: the dense weight gradient dW = dY^T X of the exact
// weight-dropout mode straight from the ROW-MAJOR activations (At = dY [K = tokens, M = out], Bt = X [tokens, N = in]),
// split over `nslab` row ranges.  Both MFMA operands have their k axis along the image ROWS, so a K step stages 32 rows x
// 128 columns of each operand (rows of 256 B: whole cache lines, one-KiB LDS-DMA pieces of 4 rows) and every fragment is
// two transposing LDS reads (ds_read_b64_tr_b16: a lane receives 8 consecutive k of its column).  Image swizzle: the 32-byte
// chunk c of row r sits at chunk c ^ (r & 7) ^ (((r >> 3) & 1) << 2) -- the two 16-lane groups of a half wave read rows 8 fq ..
// 8 fq + 3 of one logical chunk and land on eight different physical chunks, 256 B = all 64 banks.  Replaces two
// activation-sized transposes (16-byte tiles through LDS, a read and a write of every activation) plus their pad memsets per
// linear: 143 us per block at the headline shape.
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int tn_swz(int r) { return (r & 7) ^ (((r >> 3) & 1) << 2); }

__global__ __launch_bounds__(256, 4) void gemm_tn_kernel(const bf16* __restrict__ At, const int lda, const bf16* __restrict__ Bt, const int ldb,
                                                         const cara_gemm_args p, const int K, const int kslab, const size_t slab_stride,
                                                         const int tiles_n, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = 32 * 256, SLOT = 2 * IMG;   // one operand image: 32 rows x 128 columns bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int k_begin = blockIdx.y * kslab;
  const int k_end = k_begin + kslab < K ? k_begin + kslab : K;
  // staging: wave w issues pieces w, w + 4 of each image (piece = 4 rows); lane -> row l >> 4 of the piece, physical 16-byte
  // slot l & 15; the source column comes from the swizzle (an involution)
  unsigned offA[2], offB[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = (wave + 4 * t) * 4 + (lane >> 4);
    const int pc = (lane & 15) >> 1, half = lane & 1;
    const int col = ((pc ^ tn_swz(row)) << 4) + half * 8;
    offA[t] = (unsigned)row * (unsigned)(lda * 2) + (unsigned)((m0 + col) * 2);
    offB[t] = (unsigned)row * (unsigned)(ldb * 2) + (unsigned)((n0 + col) * 2);
  }
  auto stage = [&](int k0, char* dst) {
    // rows beyond k_end - 1 are clamped (their products are masked out of the A fragments below)
    const char* a = reinterpret_cast<const char*>(At);
    const char* b = reinterpret_cast<const char*>(Bt);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = (wave + 4 * t) * 4 + (lane >> 4);
      const int over = k0 + row - (k_end - 1);
      const size_t ka = (size_t)(over > 0 ? k0 - over : k0);      // per-lane clamp of the row index to k_end - 1
      glds16(a + ka * (size_t)(lda * 2) + offA[t], dst + (uwave + 4 * t) * 1024);
      glds16(b + ka * (size_t)(ldb * 2) + offB[t], dst + IMG + (uwave + 4 * t) * 1024);
    }
  };
  // fragment addresses inside an image: row 8 fq + (fr >> 2) (+ 4 for the second read), logical 32-byte chunk = the 16-column
  // tile index, 8-byte piece fr & 3
  int oa_lo[4], oa_hi[4], ob_lo[4], ob_hi[4];   // this wave's four 16-column tiles of the A image (wr) and of the B image (wc)
  {
    const int rlo = fq * 8 + (fr >> 2), rhi = rlo + 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ca = wr * 4 + i, cb = wc * 4 + i;
      oa_lo[i] = rlo * 256 + ((ca ^ tn_swz(rlo)) << 5) + (fr & 3) * 8;
      oa_hi[i] = rhi * 256 + ((ca ^ tn_swz(rhi)) << 5) + (fr & 3) * 8;
      ob_lo[i] = rlo * 256 + ((cb ^ tn_swz(rlo)) << 5) + (fr & 3) * 8;
      ob_hi[i] = rhi * 256 + ((cb ^ tn_swz(rhi)) << 5) + (fr & 3) * 8;
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (k_end - k_begin + 31) / 32;
  if (nk > 0) stage(k_begin, smem);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* sA = smem + cur * SLOT;
    const char* sB = sA + IMG;
    if (kt + 1 < nk) stage(k_begin + (kt + 1) * 32, smem + (cur ^ 1) * SLOT);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sA + oa_lo[i]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sA + oa_hi[i]));
      const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
      a[i] = bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sB + ob_lo[j]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(sB + ob_hi[j]));
      const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
      b[j] = bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
    }
    const int kbase = k_begin + kt * 32;
    if (kbase + 32 > k_end) {   // last, partial step of the slab (wave-uniform): rows >= k_end contribute nothing
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          if (kbase + fq * 8 + jj >= k_end) a[i][jj] = (bf16)0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    cur ^= 1;
  }
  // epilogue: two 32-row halves through a wave-private [32][64] fp32 image, 16-byte stores
  __syncthreads();
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  const size_t coff = (size_t)blockIdx.y * slab_stride;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[half * 2 + i][j][r];
    asm volatile("" ::: "memory");
    epilogue_rows<CARA_EPI_F32, 32>(p, stg, m0 + wr * 64 + half * 32, n0 + wc * 64, lane, coff);
    asm volatile("" ::: "memory");
  }
}
}  // namespace

extern "C" int cara_gemm_tn_f32(const void* At, int lda, const void* Bt, int ldb, float* C, int ldc, int M, int N, int K, int nslab,
                                size_t slab_stride, void* stream) {
  if (!At || !Bt || !C || M <= 0 || N <= 0 || K <= 0 || nslab <= 0 || nslab > 65535) return CARA_E_ARG;
  if ((M % 128) || (N % 128) || lda < M || ldb < N || (lda & 7) || (ldb & 7) || ldc < N || (ldc & 3)) return CARA_E_ARG;
  if ((unsigned long long)K * lda * 2 >= (1ull << 32) || (unsigned long long)K * ldb * 2 >= (1ull << 32)) return CARA_E_ARG;
  if (nslab > 1 && slab_stride < (size_t)M * ldc) return CARA_E_ARG;
  const int kslab = ((K + nslab - 1) / nslab + 31) / 32 * 32;   // rows per slab, a multiple of the K step
  if ((long long)kslab * (nslab - 1) >= K) return CARA_E_ARG;     // every slab must hold at least one row
  cara_gemm_args p = {};
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.epi = CARA_EPI_F32;
  const int tiles_n = N / 128, nwg = (M / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(nwg, nslab), dim3(256), 2 * 2 * 32 * 256, static_cast<hipStream_t>(stream),
                     static_cast<const bf16*>(At), lda, static_cast<const bf16*>(Bt), ldb, p, K, kslab, slab_stride, tiles_n, nwg);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

static int gemm_bf16_impl(const cara_gemm_args* a, void* stream, const TsPair* ts);
// measurement tools only (tools/gemm8_bench.py): pick the tile family per call sequence inside one process -- 160 / 256: that tile for
// every product it takes; 0: never; -1: the policy below decides (the default)
static int g_gemm8_override = -1;
// (test hooks, not in include/cara_hip.h: mutable process globals -- they do nothing unless the process opted in with
// CARA_ALLOW_DEBUG_SETTERS=1, which tests/conftest.py sets)
bool cara_debug_setters_allowed() {
  static const bool ok = [] { const char* e = getenv("CARA_ALLOW_DEBUG_SETTERS"); return e && atoi(e) == 1; }();
  return ok;
}
extern "C" int cara_debug_set_gemm8(int mt) {
  if (!cara_debug_setters_allowed()) return CARA_E_ARG;
  g_gemm8_override = mt;
  return CARA_OK;
}
// Which products run on the 160 x 256 x 64 tile (same-box A/Bs of the step, profiles/r04_b_*): the long-K, narrow-N ones --
//   * fc2 forward (K = 4 dim, N = dim, the adapter inside: 86 -> 74 us) and qkv dX (K = 3 dim, with its riding products as
//     workgroups behind the tiles: 64.7 -> 60.3 us);
//   * NOT a dX launch that carries long riders: fc1 dX carries 154 MB of transposed skinny products (its dVs reads dH, fc2's dU
//     reads h) that the 128 x 128 x 32 kernel streams under its four resident workgroups per CU (87 us in all); one tile per CU
//     leaves them to the end (100 us) or to helper waves that cost the tile more than they hide (92 us).  riders = 1 asks for a
//     launch that carries products: K <= CARA_GEMM8_MAXK_TS (2304) then.
// CARA_GEMM8=0 turns the tile off, CARA_GEMM8_MINK / _MAXN / _MAXK_TS move the bounds (A/B runs).  Callers that lay activations
// out for a GEMM (vit.hip: K-panel-major h / dH) ask here first: the tile reads row-major operands.
bool cara_gemm8_policy(int M, int N, int K, int riders) {
  static const int on = [] { const char* e = getenv("CARA_GEMM8"); return e ? atoi(e) : 160; }();
  static const int mink = [] { const char* e = getenv("CARA_GEMM8_MINK"); return e ? atoi(e) : 2048; }();
  static const int maxn = [] { const char* e = getenv("CARA_GEMM8_MAXN"); return e ? atoi(e) : 1024; }();
  static const int maxk_ts = [] { const char* e = getenv("CARA_GEMM8_MAXK_TS"); return e ? atoi(e) : 2304; }();
  // CARA_GEMM8_ALSO (A/B runs): more shape classes on the tile -- 1: N = 3 dim, K = dim without riders (qkv forward); 2: N = K = dim
  // without riders (proj forward); 4: N = K = dim with riders (proj dX)
  static const int also = [] { const char* e = getenv("CARA_GEMM8_ALSO"); return e ? atoi(e) : 0; }();
  if (on != 160 || M < 4096 || (M % 16) || (N % 16)) return false;
  if (N <= maxn && K >= mink && (!riders || K <= maxk_ts)) return true;
  if ((also & 1) && !riders && N == 3 * K) return true;
  if ((also & 2) && !riders && N == K && N <= maxn) return true;
  if ((also & 4) && riders && N == K && N <= maxn) return true;
  if ((also & 8) && !riders && N == 4 * K) return true;   // 8: N = 4 dim, K = dim without riders (fc1 forward)
  return false;
}

extern "C" int cara_gemm_bf16(const cara_gemm_args* a, void* stream) { return gemm_bf16_impl(a, stream, nullptr); }

extern "C" int cara_gemm_with_tskinny(const cara_gemm_args* a, const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                                      const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b, int ldg,
                                      int M, int Rp, void* stream) {
  return cara_gemm_with_tskinny_r(a, Xa, ldxa, Gta, slabs_a, K1a, Xb, ldxb, Gtb, slabs_b, K1b, want_colsum_b, ldg, M, Rp, Rp, stream);
}

extern "C" int cara_gemm_with_tskinny_r(const cara_gemm_args* a, const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a,
                                        const void* Xb, int ldxb, const void* Gtb, void* slabs_b, int K1b, int want_colsum_b, int ldg,
                                        int M, int Rp, int rank, void* stream) {
  if (rank <= 0 || rank > Rp) return CARA_E_ARG;
  // (Xa == NULL: the launch carries the second product only)
  if (!(Rp == 32 || Rp == 64) || (Xa && !ts_args_ok(Xa, ldxa, Gta, ldg, slabs_a, M, K1a, Rp)) || !ts_args_ok(Xb, ldxb, Gtb, ldg, slabs_b, M, K1b, Rp)) return CARA_E_ARG;
  TsPair ts;
  ts.a = ts_problem(Xa ? Xa : Xb, Xa ? ldxa : ldxb, Xa ? Gta : Gtb, Xa ? slabs_a : slabs_b, 0, M, Xa ? K1a : K1b, Rp);
  if (!Xa) ts.a.nblk = 0;
  ts.b = ts_problem(Xb, ldxb, Gtb, slabs_b, want_colsum_b, M, K1b, Rp);
  ts.ldg = ldg; ts.M = M; ts.any_cs = want_colsum_b != 0; ts.nt = (Rp == 32 && rank <= 16 && (!a->Ut || (a->Ut_rank > 0 && a->Ut_rank <= 16))) ? 1 : Rp / 16;
  return gemm_bf16_impl(a, stream, &ts);
}

static int g8_choice(const cara_gemm_args* a, bool riders) {
  return g_gemm8_override > 0 ? g_gemm8_override : (g_gemm8_override < 0 && cara_gemm8_policy(a->M, a->N, a->K, riders ? 1 : 0) ? 160 : 0);
}
extern "C" int cara_gemm_rider_slab_format(const cara_gemm_args* a, int Rp, int rank) {
  if (!a || !(Rp == 32 || Rp == 64) || rank <= 0 || rank > Rp) return 0;
  const int g8 = g8_choice(a, true);
  if (!g8) return 0;
  const int nt = (Rp == 32 && rank <= 16 && (!a->Ut || (a->Ut_rank > 0 && a->Ut_rank <= 16))) ? 1 : Rp / 16;
  return cara_gemm8_plan(a, g8, nt) == 2 ? 1 : 0;
}

extern "C" int cara_gemm_epi_rider_chunks(const cara_gemm_args* a) {
  if (!a || a->epi != CARA_EPI_MULH || !tile160(a) || a->Ut || (a->N & 127) || (a->ldc & 7) || (a->M & 3) || a->M <= 0) return 0;
  if (g_gemm8_override > 0) return 0;
  return (a->M + er_rows() - 1) / er_rows();
}
extern "C" int cara_gemm_dv_chunks(const cara_gemm_args* a, int riders) {
  static const int on = [] { const char* e = getenv("CARA_GEMM8"); return e ? atoi(e) : 160; }();
  if (!a || !a->er_Tt || a->epi != CARA_EPI_BF16 || on != 160 || g_gemm8_override == 0) return 0;
  return cara_gemm8_plan(a, 160, riders ? 1 : 0) == 1 ? (a->M + 159) / 160 : 0;
}
extern "C" size_t cara_gemm_epi_rider_scratch_bytes(int chunks, int N) {
  if (chunks <= 0 || N <= 0 || (N & 63)) return 0;
  const size_t nblk = (size_t)chunks * (N / 64);
  return nblk * 64 * 32 * sizeof(float) + nblk * 64 * sizeof(float);   // (the column sums sit behind slabs of the full width 32, cara_tskinny_reduce*)
}

// ts != NULL: the launch also carries a pair of transposed skinny products; only the default 128 x 128 x 32 kernel can
static int gemm_bf16_impl(const cara_gemm_args* a, void* stream, const TsPair* ts) {
  if (!a || !a->A || !a->B || !a->C) return CARA_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % BK) != 0) return CARA_E_ARG;
  if ((!a->a_panels && (a->lda < a->K || (a->lda & 7))) || a->ldb < a->K || (a->ldb & 7) || a->ldc < a->N) return CARA_E_ARG;
  const bool panels = a->a_panels || a->c_panels;
  if (panels) {   // K-panel-major activations: bf16 outputs
    if (a->a_panels < 0 || a->c_panels < 0 || (a->a_panels && a->a_panels < a->M) || (a->c_panels && a->c_panels < a->M))
      return CARA_E_ARG;
    if (a->c_panels && ((a->N & 31) || !(a->epi == CARA_EPI_BF16 || a->epi == CARA_EPI_GELU || a->epi == CARA_EPI_DGELU || a->epi == CARA_EPI_GELU_DG ||
                                         a->epi == CARA_EPI_MULH)))
      return CARA_E_ARG;
    if (a->batch > 1 || a->M <= 128) return CARA_E_ARG;
  }
  if (!(a->Rp == 0 || a->Rp == 32 || a->Rp == 64)) return CARA_E_ARG;
  if (a->Rp && ((!a->A2 && !a->Ut) || !a->B2)) return CARA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // operands are addressed with 32-bit byte offsets from the (batch-adjusted) base pointer
  const bool small_ptrs = (unsigned long long)a->M * (a->a_panels ? 32 : a->lda) * 2 < (1ull << 32) && (unsigned long long)a->N * a->ldb * 2 < (1ull << 32);
  if (!small_ptrs) return CARA_E_ARG;
  if (a->epi == CARA_EPI_RESID && (!a->aux || (a->rowscale && a->rows_per_sample <= 0))) return CARA_E_ARG;
  if ((a->epi == CARA_EPI_DGELU || a->epi == CARA_EPI_MULH) && !a->aux) return CARA_E_ARG;
  if (a->epi == CARA_EPI_MULH && a->bias) return CARA_E_ARG;
  if (ts && (a->batch > 1 || a->M <= 128 || a->B3 || (a->Ut && a->epi != CARA_EPI_BF16))) return CARA_E_ARG;
  if (a->B3 && (a->Bp || a->Ut || a->batch > 1 || a->a_panels)) return CARA_E_ARG;
  if (a->er_Tt && a->epi == CARA_EPI_BF16) {
    // dVs (+ dc) of the GEMM's own linear out of the A sub-buffers of its K loop: the 160 x 256 x 64 tile only (cara_gemm_dv_chunks)
    if (!cara_gemm_dv_chunks(a, ts ? 1 : 0)) return CARA_E_ARG;
    cara_g8_riders rd;
    if (ts) {
      auto cp = [](const TsProblem& t) { return cara_g8_product{t.X, t.Gt, t.slabs, t.cs_slabs, t.ldx, t.K1, t.nchunks, t.nblk}; };
      rd.a = cp(ts->a); rd.b = cp(ts->b); rd.ldg = ts->ldg; rd.M = ts->M; rd.any_cs = ts->any_cs ? 1 : 0; rd.nt = ts->nt;
    }
    const int rc = cara_gemm8_launch(a, static_cast<hipStream_t>(stream), 160, ts ? &rd : nullptr);
    return rc < 0 ? CARA_E_ARG : rc;
  }
  if (a->er_Tt) {   // epilogue riders: only where cara_gemm_epi_rider_chunks() says so
    if (!cara_gemm_epi_rider_chunks(a) || !a->er_Gt || !a->er_h || !a->er_slabs_v || !a->er_slabs_u || a->er_ldg < a->M || (a->er_ldg & 3) ||
        a->er_h_panels < 0 || (a->er_h_panels && a->er_h_panels < a->M))
      return CARA_E_ARG;
    // (the riders' epilogue addresses C, aux, h, T^T and G'^T with 32-bit byte offsets)
    const unsigned long long pmax = a->c_panels > a->er_h_panels ? a->c_panels : a->er_h_panels;
    if ((unsigned long long)a->M * a->ldc * 2 >= (1ull << 32) || (unsigned long long)(a->N / 32) * pmax * 64 >= (1ull << 32) ||
        (unsigned long long)a->er_ldg * 32 >= (1ull << 32))
      return CARA_E_ARG;
    if (ts && ts->nt != 1) return CARA_E_ARG;
  }
  // The MT x 256 x 64 one-workgroup-per-CU tile (gemm8.hip) where the policy asks for it (cara_gemm8_policy) and the tile takes the product
  const int g8 = a->er_Tt ? 0 : g8_choice(a, ts != nullptr);
  if (g8) {
    cara_g8_riders rd;
    if (ts) {
      auto cp = [](const TsProblem& t) { return cara_g8_product{t.X, t.Gt, t.slabs, t.cs_slabs, t.ldx, t.K1, t.nchunks, t.nblk}; };
      rd.a = cp(ts->a); rd.b = cp(ts->b); rd.ldg = ts->ldg; rd.M = ts->M; rd.any_cs = ts->any_cs ? 1 : 0; rd.nt = ts->nt;
    }
    const int rc = cara_gemm8_launch(a, st, g8, ts ? &rd : nullptr);
    if (rc >= 0) return rc;
  }
  if (a->Ut) {   // whole adapter inside the GEMM: Rp = 32, T produced here
    if (a->A2 || !a->B2 || !(a->Rp == 32 || a->Rp == 64) || !a->T_out || a->batch > 1 || (a->Tt_out && (a->ldt < a->M || (a->ldt & 7)))) return CARA_E_ARG;
    switch (a->epi) {
      case CARA_EPI_BF16: return launch32ft<CARA_EPI_BF16>(a, st, ts);
      case CARA_EPI_F32: return launch32ft<CARA_EPI_F32>(a, st);
      case CARA_EPI_GELU: return launch32ft<CARA_EPI_GELU>(a, st);
      case CARA_EPI_RESID: return launch32ft<CARA_EPI_RESID>(a, st);
      case CARA_EPI_DGELU: return launch32ft<CARA_EPI_DGELU>(a, st);
      default: return CARA_E_ARG;
    }
  }
  if (a->batch > 1) {   // batched products: plain epilogues
    if (a->A2 || a->aux || a->C2 || a->Bp || !(a->epi == CARA_EPI_F32 || a->epi == CARA_EPI_BF16) || a->batch > 65535) return CARA_E_ARG;
    return a->epi == CARA_EPI_F32 ? launch32<CARA_EPI_F32>(a, st) : launch32<CARA_EPI_BF16>(a, st);
  }
  if (const int nslab = ts ? 0 : small_m_slabs(a)) {
    switch (a->epi) {
      case CARA_EPI_BF16: return launch_small_m<CARA_EPI_BF16>(a, nslab, st);
      case CARA_EPI_F32: return launch_small_m<CARA_EPI_F32>(a, nslab, st);
      case CARA_EPI_GELU: return launch_small_m<CARA_EPI_GELU>(a, nslab, st);
      case CARA_EPI_RESID: return launch_small_m<CARA_EPI_RESID>(a, nslab, st);
      case CARA_EPI_DGELU: return launch_small_m<CARA_EPI_DGELU>(a, nslab, st);
      case CARA_EPI_GELU_DG: return launch_small_m<CARA_EPI_GELU_DG>(a, nslab, st);
      case CARA_EPI_MULH: return launch_small_m<CARA_EPI_MULH>(a, nslab, st);
      default: return CARA_E_ARG;
    }
  }
  switch (a->epi) {
    case CARA_EPI_BF16: return launch32<CARA_EPI_BF16>(a, st, ts);
    case CARA_EPI_F32: return launch32<CARA_EPI_F32>(a, st, ts);
    case CARA_EPI_GELU: return launch32<CARA_EPI_GELU>(a, st, ts);
    case CARA_EPI_RESID: return launch32<CARA_EPI_RESID>(a, st, ts);
    case CARA_EPI_DGELU: return launch32<CARA_EPI_DGELU>(a, st, ts);
    case CARA_EPI_GELU_DG: return ts ? CARA_E_ARG : launch32<CARA_EPI_GELU_DG>(a, st);
    case CARA_EPI_MULH: return launch32<CARA_EPI_MULH>(a, st, ts);
    default: return CARA_E_ARG;
  }
}

// scratch for the few-row split-K path: up to 16 slabs of 128 rows x 4096 columns of fp32 partial products
extern "C" size_t cara_gemm_scratch_bytes(void) { return (size_t)16 * 128 * 4096 * sizeof(float); }
