// Stream-K bf16 MFMA GEMM for the large products of the adapted ViT block (gfx950 / MI355X).
//
//   C = A[M,K] . B[N,K]^T (+ A2[M,Rp] . B2[N,Rp]^T) -> epilogue        (same contract as gemm.hip)
//
// Why a second structure: hipBLASLt's plain GEMM on these shapes runs at ~900 TF/s where the
// 128x128 one-tile-per-workgroup kernel of gemm.hip reaches 500-640 (tools/gemm_bench.py --blas,
// DESIGN.md section 7).  Two things separate them: the per-wave tile (64x64 there: one LDS byte per
// 32 flop) and the tile count (M = 12608 gives 600 / 450 / 150 tiles of 256x256 on 256 CUs: 2.3 /
// 1.8 / 0.6 rounds).  This kernel takes both on:
//
//  * ONE persistent 512-thread workgroup per CU; 256x256 tile, BK = 64, 8 waves as 2 (M) x 4 (N),
//    wave tile 128x64 = 8x4 accumulators of v_mfma_f32_16x16x32_bf16 (128 registers);
//  * stream-K: the tiles x K-steps iteration space is cut into G equal contiguous ranges (G = CUs),
//    so every CU issues the same number of MFMAs whatever the tile count.  A range that starts
//    inside a tile writes its fp32 partial tile to caller scratch and publishes a flag; the range
//    that holds the tile's first K-step adds the partials in fixed order (bitwise reproducible)
//    and runs the epilogue.  A workgroup only ever waits for higher-numbered workgroups' FIRST
//    piece of work, so the wait cannot deadlock under any residency;
//  * K-tiles are staged as four 16-KiB half-tiles (A0 A1 B0 B1: the 64 rows / 32 columns of each
//    wave's quadrants) by 16-byte LDS-DMA into two 64-KiB buffers, about five phases ahead of their
//    use and straight across tile boundaries (the source address follows the iteration index), with
//    a counted vmcnt(8) per phase -- the loop never drains the DMA queue;
//  * a K-tile is four phases (one 64x32 quadrant of the wave tile x K = 64 = 16 MFMAs each); the two
//    wave rows run one barrier interval apart, so on every SIMD one wave issues MFMAs while its
//    partner issues the LDS reads and DMA of its next phase;
//  * the rank-R K-extension is one more K-tile whose columns >= Rp come from a zero line;
//  * accumulators are held TRANSPOSED (B fragment as the MFMA's first operand): a lane then owns
//    four consecutive columns of one row, and the epilogue stores 8 / 16 bytes per lane straight
//    from registers (no LDS round trip; the staging buffers stay live across the epilogue).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT = 128 * BK * 2;       // half-tile: 128 rows x 128 B
constexpr int KT = 4 * HT;             // [A0][A1][B0][B1]
constexpr int LDS_BYTES = 2 * KT;      // 128 KiB
constexpr int SLOT_F32 = BM * BN;      // one partial tile, fp32
constexpr int MAXG = 256;
constexpr size_t FLAG_BYTES = 4096;

__device__ __attribute__((aligned(16))) unsigned g_zero_line[16];   // never written: zeros

struct SkPlan {
  int tiles;                 // tiles_m * tiles_n
  int tiles_m, tiles_n, gw;  // tile grid; tiles are walked in column groups of gw (row index slow inside a group)
  int nk, ipt;               // K-tiles of A/B, iterations per tile (nk + extension)
  int G, R;                  // workgroups; data-parallel rounds: workgroup g computes tiles r*G + g, r < R, whole
  int tail0, tot_t, Gt, tbase, trem;   // stream-K tail: the tiles from tail0 = R*G on; their tot_t iterations are
                                       // cut into Gt <= G ranges, range g = [g*tbase + min(g, trem), ...)
  unsigned epoch;
  float* ws;                 // [G][SLOT_F32]
  unsigned* flags;           // [G]
};

// Wave-uniform cursor over one workgroup's work: R whole tiles, then its tail range (one or two pieces).
struct Cur {
  int tm, tn, kt;      // tile coordinates and K step of this iteration
  int kb, left;        // first K step of the piece it belongs to; iterations left in the piece (this one included)
  int round, tpos;     // walker: next data-parallel round, next tail iteration
};
__device__ __forceinline__ int sk_start(int g, const SkPlan& s) {
  return g < s.Gt ? g * s.tbase + (g < s.trem ? g : s.trem) : s.tot_t;
}

// order index -> tile coordinates: column groups of gw tiles (the last may be narrower), row-major inside
// a group.  32 consecutive indices (what one XCD has in flight) then form a (32/gw) x gw block of tiles
// that shares 32/gw A panels and gw B panels in that XCD's L2.
__device__ __forceinline__ void tile_coords(int idx, const SkPlan& s, int& tm, int& tn) {
  const int per_group = s.tiles_m * s.gw;
  const int gi = idx / per_group, rem = idx - gi * per_group;
  const int c0 = gi * s.gw;
  const int w = (s.tiles_n - c0) < s.gw ? (s.tiles_n - c0) : s.gw;
  tm = rem / w;
  tn = c0 + (rem - tm * w);
}
// load the next piece of workgroup gv (tail range ends at te); false when there is none
__device__ __forceinline__ bool next_piece(Cur& c, const SkPlan& s, int gv, int te) {
  if (c.round < s.R && c.round * s.G + gv < s.tiles) {
    tile_coords(c.round * s.G + gv, s, c.tm, c.tn);
    ++c.round;
    c.kt = c.kb = 0;
    c.left = s.ipt;
    return true;
  }
  if (c.tpos < te) {
    const int t = c.tpos / s.ipt;
    c.kt = c.kb = c.tpos - t * s.ipt;
    const int room = s.ipt - c.kb, want = te - c.tpos;
    c.left = room < want ? room : want;
    c.tpos += c.left;
    tile_coords(s.tail0 + t, s, c.tm, c.tn);
    return true;
  }
  c.left = 0;
  return false;
}
__device__ __forceinline__ void cur_next(Cur& c, const SkPlan& s, int gv, int te) {
  if (c.left > 1) {
    ++c.kt;
    --c.left;
  } else {
    next_piece(c, s, gv, te);
  }
}

// per-thread constants of the staging map (t = which of the wave's two 1-KiB pieces)
struct StageConst {
  int rowA[2], rowB[2];   // tile row / column of half 0 (half 1: +64 / +32)
  int chb[2];             // byte offset of the 16-byte K chunk this lane fetches (source-side swizzle)
};
__device__ __forceinline__ StageConst stage_const(int wave, int lane) {
  StageConst sc;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int lrow = (wave * 2 + t) * 8 + (lane >> 3);
    sc.rowA[t] = (lrow >> 6) * 128 + (lrow & 63);
    sc.rowB[t] = (lrow >> 5) * 64 + (lrow & 31);
    sc.chb[t] = ((lane & 7) ^ ((lrow >> 1) & 7)) * 16;
  }
  return sc;
}

// Per-thread byte offsets (row * ld + swizzled K chunk) of the rows a thread stages, for ONE half of the
// tile a cursor is on: recomputed only when that cursor enters a new piece, so a steady-state staging call
// is one scalar add (K offset) + one vector add per 1-KiB piece.
struct StageOfs {
  unsigned a[2], b[2];
};
__device__ __forceinline__ void stage_ofs(StageOfs& o, const cara_gemm_args& p, const Cur& c, const int half, const StageConst& sc) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int ga = c.tm * 256 + half * 64 + sc.rowA[t], gb = c.tn * 256 + half * 32 + sc.rowB[t];
    ga = ga < p.M - 1 ? ga : p.M - 1;
    gb = gb < p.N - 1 ? gb : p.N - 1;
    o.a[t] = __umul24((unsigned)ga, (unsigned)p.lda * 2u) + (unsigned)sc.chb[t];
    o.b[t] = __umul24((unsigned)gb, (unsigned)p.ldb * 2u) + (unsigned)sc.chb[t];
  }
}

// Half-tile H (0 A0, 1 A1, 2 B0, 3 B1) of the iteration at `c` into K-tile buffer `buf` (`o` holds the
// offsets of that half of c's tile).  `wave` is wave-uniform (SGPR), so the LDS destination is scalar.
template <int H>
__device__ __forceinline__ void stage_half(const cara_gemm_args& p, const SkPlan& s, const Cur& c, const StageOfs& o, char* buf,
                                           const int wave, const StageConst& sc) {
  constexpr int OPB = H >> 1, HALF = H & 1;
  char* dst = buf + H * HT + wave * 2048;
  if (c.kt != s.nk) {
    const char* P = static_cast<const char*>(OPB ? p.B : p.A);
    const unsigned kb = (unsigned)c.kt * (BK * 2);
#pragma unroll
    for (int t = 0; t < 2; ++t) glds16(P + ((OPB ? o.b[t] : o.a[t]) + kb), dst + t * 1024);
  } else {
    // rank extension (one iteration per tile): [rows, Rp] operands, K chunks at or beyond Rp come from the zero line
    const bf16* P = static_cast<const bf16*>(OPB ? p.B2 : p.A2);
    const int rmax = (OPB ? p.N : p.M) - 1;
    const int r0 = (OPB ? c.tn : c.tm) * 256 + HALF * (OPB ? 32 : 64);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int gr = r0 + (OPB ? sc.rowB[t] : sc.rowA[t]);
      gr = gr < rmax ? gr : rmax;
      const int k = sc.chb[t] >> 1;
      const bf16* src = P + (size_t)gr * p.Rp + k;
      if (k >= p.Rp) src = reinterpret_cast<const bf16*>(g_zero_line);
      glds16(src, dst + t * 1024);
    }
  }
}

struct Frags {
  bf16x8 a[4][2], b0[2][2], b1[2][2];
};

__device__ __forceinline__ void read_a(Frags& f, const char* half, int aoff, int c0) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f.a[m][0] = *reinterpret_cast<const bf16x8*>(half + aoff + m * 2048 + c0);
    f.a[m][1] = *reinterpret_cast<const bf16x8*>(half + aoff + m * 2048 + (c0 ^ 64));
  }
}
__device__ __forceinline__ void read_b(bf16x8 (&b)[2][2], const char* half, int boff, int c0) {
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    b[n][0] = *reinterpret_cast<const bf16x8*>(half + boff + n * 2048 + c0);
    b[n][1] = *reinterpret_cast<const bf16x8*>(half + boff + n * 2048 + (c0 ^ 64));
  }
}

// 16 MFMAs: quadrant (MH, NH) of the wave tile over K = 64.  B fragment first: acc holds C^T tiles,
// lane (fr, fq) owns C[row fr][columns fq*4 .. fq*4+3] of each 16x16 tile.
template <int MH, int NH>
__device__ __forceinline__ void mma_quadrant(f32x4 (&acc)[8][4], const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[MH * 4 + m][NH * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n][ks], a[m][ks], acc[MH * 4 + m][NH * 2 + n], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_wait(const bool steady) {
  if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One K-tile = four phases.  has2: two more iterations follow, so every phase stages a half-tile and
// waits with vmcnt(8); otherwise the waits drain.
__device__ __forceinline__ void k_tile(f32x4 (&acc)[8][4], Frags& f, const cara_gemm_args& p, const SkPlan& s, const Cur& cur1,
                                       const Cur& cur2, const StageOfs& o1, const StageOfs& o2, char* buf, char* nbuf,
                                       const bool has1, const bool has2, const int wave, const int aoff, const int boff, const int c0,
                                       const StageConst& sc) {
  // phase 0: quadrant (0,0); B1 of l+1
  read_b(f.b0, buf + 2 * HT, boff, c0);
  __builtin_amdgcn_sched_barrier(0);
  read_a(f, buf, aoff, c0);
  if (has1) stage_half<3>(p, s, cur1, o1, nbuf, wave, sc);
  dma_wait(has2);
  wg_barrier();
  mma_quadrant<0, 0>(acc, f.a, f.b0);
  wg_barrier();
  // phase 1: quadrant (0,1); A1 of l+1
  read_b(f.b1, buf + 3 * HT, boff, c0);
  if (has1) stage_half<1>(p, s, cur1, o1, nbuf, wave, sc);
  dma_wait(has2);
  wg_barrier();
  mma_quadrant<0, 1>(acc, f.a, f.b1);
  wg_barrier();
  // phase 2: quadrant (1,1); A0 of l+2 over A0 of l (last read two phases ago)
  read_a(f, buf + HT, aoff, c0);
  if (has2) stage_half<0>(p, s, cur2, o2, buf, wave, sc);
  dma_wait(has2);
  wg_barrier();
  mma_quadrant<1, 1>(acc, f.a, f.b1);
  wg_barrier();
  // phase 3: quadrant (1,0), B0 still in registers; B0 of l+2
  if (has2) stage_half<2>(p, s, cur2, o2, buf, wave, sc);
  dma_wait(has2);
  wg_barrier();
  mma_quadrant<1, 0>(acc, f.a, f.b0);
  wg_barrier();
}

// ---- epilogue straight from the transposed accumulators ---------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue_direct(const cara_gemm_args& p, const f32x4 (&acc)[8][4], const int m0w, const int n0w,
                                                const int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const int nc = n0w + fq * 4;
  f32x4 bv[4];
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    const int n = nc + nj * 16;
    bv[nj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[nj][k] = (n + k < p.N) ? p.bias[n + k] : 0.f;
    }
  }
  const bool ldvec = (p.ldc & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0w + mi * 16 + fr;
    if (m >= p.M) continue;
    float rs = 1.f;
    if constexpr (EPI == CARA_EPI_RESID) rs = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.f;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      const int n = nc + nj * 16;
      if (n >= p.N) continue;
      const bool vec = ldvec && (n + 4 <= p.N);
      const size_t o = (size_t)m * p.ldc + n;
      f32x4 v = acc[mi][nj] + bv[nj];
      if constexpr (EPI == CARA_EPI_F32 || EPI == CARA_EPI_RESID) {
        if constexpr (EPI == CARA_EPI_RESID) {
          const float* xin = static_cast<const float*>(p.aux) + o;
          if (vec) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(xin);
            v = x + rs * v;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (n + k < p.N) ? xin[k] + rs * v[k] : 0.f;
          }
        }
        float* dst = static_cast<float*>(p.C) + o;
        if (vec) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (n + k < p.N) dst[k] = v[k];
        }
      } else {
        bf16x4 out, out2;
        if constexpr (EPI == CARA_EPI_BF16) {
#pragma unroll
          for (int k = 0; k < 4; ++k) out[k] = (bf16)v[k];
        } else if constexpr (EPI == CARA_EPI_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            out2[k] = (bf16)v[k];
            out[k] = (bf16)gelu_erf(v[k]);
          }
        } else {  // CARA_EPI_DGELU
          const bf16* up = static_cast<const bf16*>(p.aux) + o;
          bf16x4 u;
          if (vec) {
            u = *reinterpret_cast<const bf16x4*>(up);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = (n + k < p.N) ? up[k] : (bf16)0.f;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) out[k] = (bf16)(v[k] * gelu_erf_grad((float)u[k]));
        }
        bf16* dst = static_cast<bf16*>(p.C) + o;
        if (vec) {
          *reinterpret_cast<bf16x4*>(dst) = out;
          if constexpr (EPI == CARA_EPI_GELU) *reinterpret_cast<bf16x4*>(static_cast<bf16*>(p.C2) + o) = out2;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (n + k < p.N) {
              dst[k] = out[k];
              if constexpr (EPI == CARA_EPI_GELU) (static_cast<bf16*>(p.C2) + o)[k] = out2[k];
            }
        }
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_sk_kernel(const cara_gemm_args p, const SkPlan s) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  // workgroups b, b+8, ... share an XCD: give them consecutive logical indices, so that in every round an XCD
  // computes 32 consecutive tiles of the grouped order, and the partial tile of range g+1 is read by range g
  // on the same XCD
  const int gv = xcd_remap(blockIdx.x, s.G);
  const int tb = sk_start(gv, s), te = sk_start(gv + 1, s);
  int rounds = s.R;
  if (rounds > 0 && (rounds - 1) * s.G + gv >= s.tiles) --rounds;   // the last round may be partly filled
  const int n_it = rounds * s.ipt + (te - tb);
  if (n_it <= 0) return;   // workgroup-uniform

  const int aoff = (wr * 64 + fr) * 128, boff = (wc * 32 + fr) * 128;
  const int c0 = (fq ^ (fr >> 1)) << 4;
  const StageConst sc = stage_const(wave, lane);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cur cur, cur1, cur2;       // the iteration being computed, the next one, the one after
  cur.round = 0;
  cur.tpos = tb;
  next_piece(cur, s, gv, te);
  cur1 = cur;
  cur_next(cur1, s, gv, te);
  cur2 = cur1;
  cur_next(cur2, s, gv, te);
  StageOfs o1, o2;           // half 1 of cur1's tile (B1, A1 are staged one iteration ahead), half 0 of cur2's (A0, B0: two ahead)

  // prologue: iteration 0 whole, A0/B0 of iteration 1 (issue order A0 B0 B1 A1 per iteration, as in the loop)
  stage_ofs(o2, p, cur, 0, sc);
  stage_ofs(o1, p, cur, 1, sc);
  stage_half<0>(p, s, cur, o2, smem, wave, sc);
  stage_half<2>(p, s, cur, o2, smem, wave, sc);
  stage_half<3>(p, s, cur, o1, smem, wave, sc);
  stage_half<1>(p, s, cur, o1, smem, wave, sc);
  if (n_it > 1) {
    stage_ofs(o2, p, cur1, 0, sc);
    stage_half<0>(p, s, cur1, o2, smem + KT, wave, sc);
    stage_half<2>(p, s, cur1, o2, smem + KT, wave, sc);
  }
  dma_wait(n_it > 1);   // A0, B0 of iteration 0 landed (4 younger half-tiles may fly)
  stage_ofs(o1, p, cur1, 1, sc);
  stage_ofs(o2, p, cur2, 0, sc);
  wg_barrier();
  if (wr == 1) wg_barrier();   // the second wave row runs one barrier interval behind the first

  Frags f;
  for (int l = 0; l < n_it; ++l) {
    char* buf = smem + (l & 1) * KT;
    char* nbuf = smem + ((l + 1) & 1) * KT;
    const bool has1 = l + 1 < n_it;
    k_tile(acc, f, p, s, cur1, cur2, o1, o2, buf, nbuf, has1, l + 2 < n_it, wave, aoff, boff, c0, sc);

    if (cur.left == 1) {
      // ---- end of a piece: K steps [cur.kb, cur.kt] of tile (cur.tm, cur.tn) ----
      const bool tile_end = cur.kt == s.ipt - 1;
      if (wr == 0) wg_barrier();   // bring the two wave rows level
      if (cur.kb != 0) {
        // not the tile's first K-step: publish the partial sums (plain stores, ONE agent-scope release)
        float* slot = s.ws + (size_t)gv * SLOT_F32 + (size_t)(wave * 32) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(slot + (i * 4 + j) * 256) = acc[i][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(s.flags + gv, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (!tile_end) {
          // holder of the first K-step: add the partial tiles of the ranges that continue this tile,
          // in range order.  They were each range's FIRST work, so they are normally long there.
          // (a piece that stops short of its tile's end is the last of this workgroup: cur.tpos == te)
          const int tile_last = (te / s.ipt + 1) * s.ipt;
          for (int g2 = gv + 1; g2 < s.G && sk_start(g2, s) < tile_last; ++g2) {
            if (sk_start(g2 + 1, s) == sk_start(g2, s)) continue;   // empty range
            if (tid == 0) {
              while (__hip_atomic_load(s.flags + g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != s.epoch)
                __builtin_amdgcn_s_sleep(8);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const float* slot = s.ws + (size_t)g2 * SLOT_F32 + (size_t)(wave * 32) * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(slot + (i * 4 + j) * 256);
          }
        }
        epilogue_direct<EPI>(p, acc, cur.tm * BM + wr * 128, cur.tn * BN + wc * 64, lane);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has1 && wr == 1) wg_barrier();   // stagger again
    }
    cur = cur1;
    cur1 = cur2;
    cur_next(cur2, s, gv, te);
    if (cur1.kt == cur1.kb) stage_ofs(o1, p, cur1, 1, sc);   // entered a new piece (offsets of a dead cursor are never used)
    if (cur2.kt == cur2.kb) stage_ofs(o2, p, cur2, 0, sc);
  }
}

int g_num_cu = 0;
unsigned g_epoch = 0x5ca1ab00u;
long g_launches = 0;

template <int EPI>
int launch_sk(const cara_gemm_args* a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sk_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return CARA_E_LAUNCH;
    attr_set = true;
  }
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return CARA_E_LAUNCH;
    g_num_cu = n < MAXG ? n : MAXG;
  }
  SkPlan s;
  s.tiles_m = (a->M + BM - 1) / BM;
  s.tiles_n = (a->N + BN - 1) / BN;
  const int ngroups = (s.tiles_n + 7) / 8;
  s.gw = (s.tiles_n + ngroups - 1) / ngroups;   // column groups of at most 8 tiles, as even as possible
  s.nk = a->K / BK;
  s.ipt = s.nk + (a->Rp > 0 ? 1 : 0);
  const int tiles = s.tiles_m * s.tiles_n;
  static int forced_g = -1;
  if (forced_g < 0) {
    const char* e = getenv("CARA_GEMM_SK_GRID");
    forced_g = e ? atoi(e) : 0;
  }
  const int G = (forced_g > 0 && forced_g <= MAXG) ? forced_g : g_num_cu;
  s.G = G;
  s.tiles = tiles;
  // Tail policy.  Cutting the leftover tiles (tiles mod G) into G equal K ranges balances the MFMA work, but
  // every range boundary costs one 256-KiB fp32 partial tile written, published and read back (measured
  // 15-30 us per launch on MI355X) -- more than the idle CUs of a partly filled round cost unless a tile is
  // long (>= sk_min_ipt K-steps; CARA_GEMM_SK_TAIL overrides, 0 = never).
  const char* e_tail = getenv("CARA_GEMM_SK_TAIL");   // read per launch: tests switch it
  const int sk_min_ipt = e_tail ? atoi(e_tail) : 96;
  const bool sk_tail = sk_min_ipt > 0 && s.ipt >= sk_min_ipt && tiles % G != 0;
  s.R = sk_tail ? tiles / G : (tiles + G - 1) / G;
  s.tail0 = sk_tail ? s.R * G : tiles;
  const int tot_t = (tiles - s.tail0) * s.ipt;
  // at least 4 K-steps per tail range (a tile cut into many short ranges costs more in partial-tile traffic
  // than it gains in balance)
  int Gt = tot_t / 4 < G ? tot_t / 4 : G;
  if (Gt < 1) Gt = 1;
  s.Gt = Gt;
  s.tot_t = tot_t;
  s.tbase = tot_t / Gt;
  s.trem = tot_t % Gt;
  s.epoch = ++g_epoch;
  s.flags = static_cast<unsigned*>(a->scratch);
  s.ws = reinterpret_cast<float*>(static_cast<char*>(a->scratch) + FLAG_BYTES);
  hipLaunchKernelGGL(gemm_sk_kernel<EPI>, dim3(G), dim3(512), LDS_BYTES, st, *a, s);
  CARA_CHECK_LAUNCH();
  ++g_launches;
  return CARA_OK;
}

}  // namespace

extern "C" long cara_debug_gemm_persistent_launches(void) { return g_launches; }
extern "C" size_t cara_gemm_scratch_bytes(void) { return FLAG_BYTES + (size_t)MAXG * SLOT_F32 * sizeof(float); }

// internal entry used by cara_gemm_bf16 (arguments validated there); CARA_E_ARG = shape not taken
int cara_gemm_sk_dispatch(const cara_gemm_args* a, hipStream_t st) {
  if (!a->scratch || a->scratch_bytes < cara_gemm_scratch_bytes()) return CARA_E_ARG;
  switch (a->epi) {
    case CARA_EPI_BF16: return launch_sk<CARA_EPI_BF16>(a, st);
    case CARA_EPI_F32: return launch_sk<CARA_EPI_F32>(a, st);
    case CARA_EPI_GELU: return launch_sk<CARA_EPI_GELU>(a, st);
    case CARA_EPI_RESID: return launch_sk<CARA_EPI_RESID>(a, st);
    case CARA_EPI_DGELU: return launch_sk<CARA_EPI_DGELU>(a, st);
    default: return CARA_E_ARG;
  }
}
