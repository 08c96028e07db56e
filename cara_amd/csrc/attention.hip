// Fused attention forward / backward for the ViT token counts (head dim 64) on gfx950: N <= 224 (ViT-B/16
// @224: 197 tokens) on the register-resident path described below, N <= 608 (ViT-L/16 @384: 577 tokens) on
// the same kernels with the whole K/V (or Q/dO) of a head still in LDS (152 KiB, one workgroup per CU) and a
// forward that sweeps the key tiles twice (max + sum, then P.V) instead of holding the score rows.
// Reference semantics: /root/reference/src/cara/cara.py:43-48
//     attn = softmax((q @ k^T) * scale); x = (attn @ v).transpose(1, 2).reshape(B, N, C)
// with q, k, v the three [B,H,N,64] views of the qkv activation laid out exactly as cara.py:39
// reshapes it ([B*N, 3*H*64], column k*H*64 + h*64 + d).  attn_drop has p = 0 in the reference
// configuration (timm attn_drop_rate = 0), so no dropout on the probabilities.
//
// Forward: the whole K/V of one head sits in LDS (28 KiB + 33 KiB), each wave owns 32 query rows
// and, because N is small, the whole score row block in registers (7 tiles of 32x32): exact
// two-pass softmax, no online rescaling.  Scores are computed TRANSPOSED (S^T = K Q^T) with
// v_mfma_f32_32x32x16_bf16 so that a query's row lives in ONE lane (max / sum = in-lane reduction
// + one cross-half shuffle) and the probability accumulators are already the A operand of P.V
// (no LDS round trip).  K and V sit in LDS as plain row-major (swizzled) images; the P.V B fragments
// (keys on the k axis) come out of the V image with the transposing LDS read ds_read_b64_tr_b16.
//
// Backward: two kernels, both recomputing P from Q, K and the forward's LSE.  (1) dK/dV: a
// workgroup of 7 waves per (batch, head); wave w owns keys 32w..32w+31 and keeps dK^T, dV^T in
// accumulators while sweeping the query tiles; S and dP are computed with the key on the lane, so
// P and dS feed dV^T += dO^T P and dK^T += Q^T dS straight from the accumulators.  (2) dQ: the
// forward's structure (wave = 32 queries, query on the lane): dS^T feeds dQ += dS K from the
// accumulators.  No atomics, no cross-wave sums: results are bitwise reproducible.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr int NMAX = 224;     // 7 tiles of 32: the score rows of a wave fit in registers
constexpr int NMAX_LONG = 608;  // 19 tiles: two [608][64] bf16 images + the row constants are 160 512 B of the 160 KiB

// Swizzle key of a row of a [rows][64] bf16 image (128-byte rows, eight 16-byte chunks): the BIT-REVERSED (row >> 1) & 7.
// ds_read_b128 row fragments need the 8 even (and the 8 odd) rows of a 16-lane service group on different chunks: any
// bijection of (row >> 1) & 7 does that.  The transposing reads (ds_read_b64_tr_b16: 32 lanes = rows R .. R + 3, four
// consecutive chunks each) also need rows R and R + 2 -- the same 256-byte bank window -- on different 64-byte HALVES of their
// rows: with the plain key their keys differ in bit 0 only and the two rows collide (2-way conflict on every transposing
// read: SQ_LDS_BANK_CONFLICT was 26 % of SQ_LDS_IDX_ACTIVE); bit-reversed, keys of rows two apart always differ in bit 2.
__device__ __forceinline__ int swzk(int row) {
  const int t = (row >> 1) & 7;
  return ((t & 1) << 2) | (t & 2) | (t >> 2);
}
__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ swzk(row)) << 4); }
// row of the 32x32 C/D layout held in register r of lane half h
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
  return o;
}

// Two score elements per vector instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process a register PAIR at the rate of one
// scalar-operand instruction; v_exp_f32 has no packed form).  Written out with two-element vectors on ALIGNED pairs (registers 2i,
// 2i + 1 of an accumulator): left to itself hipcc paired the multiplies of the backward's dS one register off and then moved the
// packed halves back into place with v_mov / v_alignbit / v_perm -- 90 vector instructions per 32 x 32 tile where 48 do (r05).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
__device__ __forceinline__ unsigned pk16(const f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 exp2_2(const f32x2 a) { return f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ bf16x8 dwords8(const unsigned (&w)[8], const int s) {   // = pack8 of the converted values: elements 8 s .. 8 s + 7
  const u32x4_ v = {w[4 * s], w[4 * s + 1], w[4 * s + 2], w[4 * s + 3]};
  return __builtin_bit_cast(bf16x8, v);
}

// B/A fragment whose k axis runs over image ROWS: element j of lane (c = lane & 31, h = lane >> 5) is
// img[base + 8*(j>>2) + 4h + (j&3)][cbase + c] -- the k order in which the 32x32x16 accumulator of a
// previous product is consumed as an operand (register 8s+j of lane half h = row 16s + 8(j>>2) + 4h
// + (j&3)).  Taken from a ROW-MAJOR swizzled [rows][64] image (rows of 128 B, chunk XOR of
// swz128) with the CDNA4 transposing LDS read.  ds_read_b64_tr_b16 works per group of 16 lanes:
// lane 4q+p of a group supplies the address of (row q, columns 4p..4p+3) of a 4 x 16 block and lane
// i receives column i of the four rows.  Group g = lane >> 4 covers columns cbase + 16*(g&1) ..+15,
// rows base + 4*(g>>1) .. +3 (first read) and +8 (second read).  All 64 lanes must be active.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_frag_rm(const char* img, int cbase, int base, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int row = base + 4 * (g >> 1) + (i >> 2);
  const int chunk = ((cbase + 16 * (g & 1)) >> 3) + ((i & 3) >> 1);
  const int sub = (i & 1) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + swz128(row, chunk) + sub));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + swz128(row + 8, chunk) + sub));
  // (bit-cast the whole vectors: __builtin_bit_cast of a single vector ELEMENT reads element 0)
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 o = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  return o;
}

// npad = N rounded up to a multiple of 32: the images hold npad rows (rows >= N duplicate row N-1)
// Lane-constant byte offsets inside a 32-row block of a swizzled [rows][64] image (row stride 128 B): the swizzle
// term swzk(row) only depends on the row's position inside its 32-row block, so the per-tile address of every
// fragment is `block * 4096 + constant` -- computed once per kernel instead of ~10 VALU instructions per read
// (rocprofv3 counted 2 300-2 450 VALU instructions per wave in these kernels, most of them address arithmetic
// and mask selects: the matrix pipe was busy 8 % of a wave's life).
struct RowOfs {   // ds_read_b128 row fragments: lane (ql = lane & 31, h = lane >> 5), K sub-step ks
  int o[4];
};
__device__ __forceinline__ RowOfs row_ofs(int lane) {
  const int ql = lane & 31, h = lane >> 5;
  RowOfs r;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) r.o[ks] = ql * 128 + (((ks * 2 + h) ^ swzk(ql)) << 4);
  return r;
}
struct TrOfs {    // the two ds_read_b64_tr_b16 of tr_frag_rm(img, cbase = dt*32, base = block*32 + st*16)
  int lo[2][2], hi[2][2];
};
__device__ __forceinline__ TrOfs tr_ofs(int lane) {
  const int i = lane & 15, g = lane >> 4;
  TrOfs t;
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int rl = st * 16 + 4 * (g >> 1) + (i >> 2);
      const int chunk = ((dt * 32 + 16 * (g & 1)) >> 3) + ((i & 3) >> 1);
      const int sub = (i & 1) * 8;
      t.lo[st][dt] = rl * 128 + ((chunk ^ swzk(rl)) << 4) + sub;
      t.hi[st][dt] = (rl + 8) * 128 + ((chunk ^ swzk(rl + 8)) << 4) + sub;
    }
  return t;
}
__device__ __forceinline__ bf16x8 tr_frag_at(const char* blk, int lo, int hi) {
  const s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(blk + lo));
  const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(blk + hi));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, l), h4 = __builtin_bit_cast(bf16x4, u);
  bf16x8 o = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  return o;
}

__device__ __forceinline__ void stage_rows_swz(const bf16* __restrict__ src, int ld, int N, char* img, int tid, int nthreads,
                                               int npad) {
  for (int idx = tid; idx < npad * 8; idx += nthreads) {
    const int n = idx >> 3, c = idx & 7;
    const int nn = n < N ? n : N - 1;
    *reinterpret_cast<uint4*>(img + swz128(n, c)) = *reinterpret_cast<const uint4*>(src + (size_t)nn * ld + c * 8);
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// LDS: K and V, both as swizzled row-major images of npad rows

template <int NW>  // waves per workgroup: 7 covers N <= 224 with ONE staging of K/V per head
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                              float* __restrict__ lse, int N, int H, float scale, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + npad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
  const int ld = 3 * H * HD;
  const bf16* qb = qkv + (size_t)b * N * ld + head * HD;
  const bf16* kb = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;

  stage_rows_swz(kb, ld, N, Ks, tid, NW * 64, npad);
  stage_rows_swz(vb, ld, N, Vs, tid, NW * 64, npad);
  __syncthreads();

  const int q0 = (blockIdx.y * NW + wave) * 32;
  if (q0 >= N) return;
  const int ql = lane & 31, h = lane >> 5;
  const int qrow = (q0 + ql) < N ? (q0 + ql) : N - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qrow * ld + ks * 16 + h * 8);

  const int nkt = (N + 31) >> 5;
  f32x16 s[7];
#pragma unroll
  for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    if (kt < nkt) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + swz128(kt * 32 + ql, ks * 2 + h));
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kt], 0, 0, 0);
      }
    }
  }
  // S^T layout: column (lane & 31) = query, row = key kt*32 + crow(r, h)
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < 7; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + crow(r, h);
      s[kt][r] = key < N ? s[kt][r] : -3.0e38f;
      mx = fmaxf(mx, s[kt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c2 = scale * 1.4426950408889634f;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 7; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2f((s[kt][r] - mx) * c2);
      s[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 7; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 pa = pack8(s[kt], st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 vf = tr_frag_rm(Vs, dt * 32, kt * 32 + st * 16, lane);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vf, o[dt], 0, 0, 0);
        }
      }
    }
  }
  // O layout: column (lane & 31) = d, row = query q0 + crow(r, h); 1/sum of that query sits in
  // lane crow(r, h)
  const float inv = 1.0f / sum;
  bf16* ob = out + (size_t)b * N * (H * HD) + head * HD;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qq = crow(r, h);
    const float iv = __shfl(inv, qq, 64);
    if (q0 + qq < N) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ob[(size_t)(q0 + qq) * (H * HD) + dt * 32 + ql] = (bf16)(o[dt][r] * iv);
    }
  }
  if (h == 0 && q0 + ql < N) lse[(size_t)bh * N + q0 + ql] = mx * scale + __logf(sum);
}

// Forward for 224 < N <= 608: same staging and fragment scheme, but the score rows no longer fit in
// registers, so the key tiles are swept twice -- pass 1 recomputes S^T tile by tile for the row maxima and
// the sums of exp, pass 2 recomputes it again and feeds P straight into P.V.  Still the exact softmax of
// the short kernel (no running rescale); QK^T is 1/3 of the work, so the second sweep costs about a third.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_long_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                   float* __restrict__ lse, int N, int H, float scale, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + npad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
  const int ld = 3 * H * HD;
  const bf16* qb = qkv + (size_t)b * N * ld + head * HD;
  const bf16* kb = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  stage_rows_swz(kb, ld, N, Ks, tid, NW * 64, npad);
  stage_rows_swz(vb, ld, N, Vs, tid, NW * 64, npad);
  __syncthreads();

  const int ql = lane & 31, h = lane >> 5;
  const int nkt = npad >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int last_keys = N - (nkt - 1) * 32;   // valid keys of the last tile (only that tile needs a mask)
  // r05: the workgroup walks the query groups of its head (group = NW x 32 queries; gridDim.y = 1 since r05) instead of one group
  // per workgroup: K / V are staged ONCE per head -- at 577 tokens three workgroups per head each staged all 148 KB, six rounds of
  // workgroups on the chip, 56 % of the wave-cycles waiting (profiles/r05_pmc_attn.txt).  No barrier inside the loop.
  for (int qg = blockIdx.y;; qg += gridDim.y) {
  const int q0 = (qg * NW + wave) * 32;
  if (q0 >= N) break;
  const int qrow = (q0 + ql) < N ? (q0 + ql) : N - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qrow * ld + ks * 16 + h * 8);

  auto score_tile = [&](int kt) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    const char* kb_ = Ks + kt * 4096;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb_ + ro.o[ks]);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], t, 0, 0, 0);
    }
    if (kt == nkt - 1) {   // S^T: row = key, column = query
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = crow(r, h) < last_keys ? t[r] : -3.0e38f;
    }
    return t;
  };
  // pass 1: row maximum, then (with the maximum known) the sum of exponentials, in the key order of pass 2
  float mx = -3.0e38f;
  for (int kt = 0; kt < nkt; ++kt) {
    const f32x16 t = score_tile(kt);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, t[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mxc = mx * c2;
  float sum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  // pass 2: P = exp(S - max) (masked keys give exp(-huge) = 0), sum and P.V together
  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 t = score_tile(kt);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      t[r] = __builtin_amdgcn_exp2f(t[r] * c2 - mxc);   // <= 0: raw v_exp_f32, no denormal fix-up code
      sum += t[r];
    }
    const char* vb_ = Vs + kt * 4096;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 pa = pack8(t, st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 vf = tr_frag_at(vb_, to.lo[st][dt], to.hi[st][dt]);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vf, o[dt], 0, 0, 0);
      }
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  bf16* ob = out + (size_t)b * N * (H * HD) + head * HD;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qq = crow(r, h);
    const float iv = __shfl(inv, qq, 64);
    if (q0 + qq < N) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ob[(size_t)(q0 + qq) * (H * HD) + dt * 32 + ql] = (bf16)(o[dt][r] * iv);
    }
  }
  if (h == 0 && q0 + ql < N) lse[(size_t)bh * N + q0 + ql] = mx * scale + __logf(sum);
  }
}

// ------------------------------------------------------------------------------------------
// Forward for 128 < N <= 224 (ViT-B/16 @224: 197 tokens), the headline path: a PERSISTENT kernel, one 7-wave
// workgroup per CU walking the (batch, head) pairs bh = block, block + grid, ... (64 x 12 = 768 pairs = exactly 3 per
// CU).  The kernel above is bound by what happens around its arithmetic: K / V are staged through registers and a
// barrier before any MFMA can issue, two workgroups per CU give 1.5 rounds of 512 slots, and the scores are computed
// twice to get under 128 VGPRs.  Here
//   * K and V of head j+1 stream into the OTHER pair of LDS images by LDS-DMA (global_load_lds, full 128-byte rows,
//     swizzle on the source address) while head j computes: counted vmcnt, raw s_barrier, nothing drains the queue;
//   * a wave's 32 query rows come the same way into a wave-private 4-KiB image, issued as soon as the wave's Q K^T of
//     the previous head has consumed its fragments;
//   * one workgroup per CU leaves a wave 256 VGPRs: the whole score row block stays in registers (7 tiles of 32 x 32),
//     one sweep, exact two-pass softmax, P feeds P.V from the accumulators as before.
// LDS: 2 x (K + V) x 28 KiB + 7 x 4 KiB = 140 KiB.
// ------------------------------------------------------------------------------------------
constexpr int PF_WAVES = 7;

// LDS-DMA piece written as inline asm (wave-uniform 64-bit base, per-lane 32-bit byte offset, LDS byte address in M0):
// hipcc does not see it, so it cannot put its own s_waitcnt vmcnt(0) in front of later LDS reads (it does that for the
// builtin whenever it cannot prove that the DMA's destination and the read do not alias, which would drain the
// next head's images in the middle of this head's P.V).  Every wait for these pieces is written by hand below.
__device__ __forceinline__ void glds16_hidden(const char* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

#ifdef CARA_ATTN_STAMPS
// Diagnostic build (tools/attn_stamps.py): wave 0 of every workgroup records s_memrealtime (100 MHz) at eight points of
// every head it walks, into a buffer of its own: [block][head slot (<= 4)][8].
__device__ unsigned long long* g_attn_stamp_buf = nullptr;
extern "C" int cara_debug_attn_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define ATTN_STAMP(i)                                                                                              \
  do {                                                                                                             \
    if (g_attn_stamp_buf && tid == 0 && slot < 4) g_attn_stamp_buf[((size_t)blockIdx.x * 4 + slot) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define ATTN_STAMP(i)
#endif
__global__ __launch_bounds__(PF_WAVES * 64, 1) void attn_fwd_persist_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                            float* __restrict__ lse, int N, int H, int BH, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPAD = 224, IMG = NPAD * 128;       // one K or V image
  char* Qs = smem + 4 * IMG;                         // wave-private Q images behind the two (K, V) pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ld = 3 * H * HD;
  const int ql = lane & 31, h = lane >> 5;
  const int q0 = wave * 32;
  const int nkt = (N + 31) >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int last_keys = N - (nkt - 1) * 32;
  char* myQ = Qs + wave * 4096;

  // per-lane 32-bit byte offsets of the DMA pieces (head-independent; the head's base pointer is wave-uniform, so a
  // piece is "SGPR base + VGPR offset" with no 64-bit vector arithmetic).  K and V of one head: 2 x 28 pieces of 8 rows,
  // wave w takes pieces w, w + 7, ... (8 per wave); Q: the wave's own 32 rows = 4 pieces.
  unsigned kvoff[8], kvdst[8], qoff[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int q = wave + t * PF_WAVES;              // 0..55: < 28 -> K, else V
    const bool isv = q >= 28;
    const int piece = isv ? q - 28 : q;
    const int row = piece * 8 + (lane >> 3);
    const int rr = row < N ? row : N - 1;
    const int c = (lane & 7) ^ swzk(row);
    kvoff[t] = (unsigned)rr * (unsigned)(ld * 2) + (unsigned)(c * 16) + (unsigned)((isv ? 2 : 1) * H * HD * 2);
    kvdst[t] = (unsigned)((isv ? IMG : 0) + piece * 1024);   // wave-uniform
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = t * 8 + (lane >> 3);
    int gr = q0 + row;
    gr = gr < N ? gr : N - 1;
    const int c = (lane & 7) ^ swzk(row);
    qoff[t] = (unsigned)gr * (unsigned)(ld * 2) + (unsigned)(c * 16);
  }
  auto lds_of = [](const char* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)p); };   // LDS byte address
  auto head_base = [&](int bh) {
    const int b = bh / H, hd = bh - b * H;
    return reinterpret_cast<const char*>(qkv + (size_t)b * N * ld + hd * HD);   // wave-uniform
  };
  auto stage_kv = [&](int bh, int buf) {
    const char* base = head_base(bh);
    char* img = smem + buf * 2 * IMG;
#pragma unroll
    for (int t = 0; t < 8; ++t) glds16_hidden(base, kvoff[t], lds_of(img) + __builtin_amdgcn_readfirstlane(kvdst[t]));
  };
  auto stage_q = [&](int bh) {
    const char* base = head_base(bh);
#pragma unroll
    for (int t = 0; t < 4; ++t) glds16_hidden(base, qoff[t], lds_of(myQ) + t * 1024);
  };

  int bh = blockIdx.x;
  if (bh >= BH) return;
  stage_kv(bh, 0);
  stage_q(bh);
  int cur = 0;
#ifdef CARA_ATTN_STAMPS
  int slot = -1;
#endif
  for (; bh < BH; bh += gridDim.x) {
    const int nxt = bh + gridDim.x;
#ifdef CARA_ATTN_STAMPS
    ++slot;
#endif
    ATTN_STAMP(0);
    // ONE barrier per head: behind it every wave's pieces of this head's K, V and Q have landed (they were issued during the
    // previous head -- everything this wave has outstanding, its output stores included, is old by now) AND every wave is
    // through with the previous head, whose K / V images the next head's pieces may therefore overwrite.  Those pieces are
    // handed out over the key tiles of the S^T loop below: issued between two barriers, as they used to be, the 56 pieces
    // of a head held all seven waves for 1.6 us while the CU's load path accepted them (time stamps, tools/attn_stamps.py).
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(1);
    const char* Ks = smem + cur * 2 * IMG;
    const char* Vs = Ks + IMG;
    const int b = bh / H, head = bh - b * H;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(myQ + ro.o[ks]);
    // S^T = K Q^T, the K fragments of tile kt + 1 requested before the MFMAs of tile kt (one wave or two per SIMD: nothing
    // else hides the LDS latency)
    f32x16 s[7];
    bf16x8 ka[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[0][ks] = *reinterpret_cast<const bf16x8*>(Ks + ro.o[ks]);
    const char* nbase = head_base(nxt < BH ? nxt : bh);
    const unsigned nimg = lds_of(smem + (cur ^ 1) * 2 * IMG);
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      if (nxt < BH && (kt & 1) == 0)   // the next head's K / V: pieces 0-3 behind the even tiles here, 4-7 in the P V loop
        glds16_hidden(nbase, kvoff[kt >> 1], nimg + __builtin_amdgcn_readfirstlane(kvdst[kt >> 1]));
      if (kt < nkt) {
        if (kt + 1 < nkt) {
          const char* kb_ = Ks + (kt + 1) * 4096;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) ka[(kt + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(kb_ + ro.o[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[kt & 1][ks], qf[ks], s[kt], 0, 0, 0);
      }
    }
    ATTN_STAMP(2);
    // the Q fragments are in registers: the wave's Q image may take the next head's rows
    if (nxt < BH) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      stage_q(nxt);
    }
    // S^T layout: column (lane & 31) = query, row = key kt*32 + crow(r, h); only the last key tile needs a mask
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
      if (kt < nkt) {
        if (kt == nkt - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = crow(r, h) < last_keys ? s[kt][r] : -3.0e38f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    ATTN_STAMP(3);
    const float mxc = mx * c2;
    float sum = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
      if (kt < nkt) {
        // the four V fragments of this key tile are requested first: the exponentials below cover their latency
        const char* vb_ = Vs + kt * 4096;
        if (nxt < BH && (kt & 1) == 0)
          glds16_hidden(nbase, kvoff[4 + (kt >> 1)], nimg + __builtin_amdgcn_readfirstlane(kvdst[4 + (kt >> 1)]));
        bf16x8 vf[2][2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vf[st][dt] = tr_frag_at(vb_, to.lo[st][dt], to.hi[st][dt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] * c2 - mxc);   // <= 0: raw v_exp_f32 (masked keys: exp(-huge) = 0)
          sum += s[kt][r];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 pa = pack8(s[kt], st);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][dt], pa, o[dt], 0, 0, 0);   // O^T = V^T P^T: d on the rows
        }
      }
    }
    if (nxt < BH) {
#pragma unroll
      for (int kt = 0; kt < 7; kt += 2)   // (the pieces of tiles this N does not have)
        if (kt >= nkt) glds16_hidden(nbase, kvoff[4 + (kt >> 1)], nimg + __builtin_amdgcn_readfirstlane(kvdst[4 + (kt >> 1)]));
    }
    sum += __shfl_xor(sum, 32, 64);
    ATTN_STAMP(4);
    const float inv = 1.0f / sum;
    // O^T layout: column (lane & 31) = QUERY, register r of lane half h = d = 32 dt + 8 (r >> 2) + 4 h + (r & 3): a lane
    // holds its query's row in runs of four d, and 1 / sum of that query is the lane's own.  Lane halves swap runs
    // (v_permlane32_swap) so that each lane ends up with two runs of EIGHT consecutive d per dt: four 16-byte stores per
    // lane and head instead of 32 two-byte ones (the store tail was 21 % of the kernel).
    bf16* ob = out + (size_t)b * N * (H * HD) + head * HD;
#ifdef CARA_ABLATE_ATTN_STORES
    if (q0 < N && inv == 123.f) {
#else
    if (q0 < N) {
#endif
      bf16* orow = ob + (size_t)(q0 + ql < N ? q0 + ql : N - 1) * (H * HD);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        unsigned w[4][2];   // run g = r >> 2 of this lane: d = 32 dt + 8 g + 4 h .. + 3, as two packed dwords
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x2 lo = {(bf16)(o[dt][4 * g] * inv), (bf16)(o[dt][4 * g + 1] * inv)};
          const bf16x2 hi = {(bf16)(o[dt][4 * g + 2] * inv), (bf16)(o[dt][4 * g + 3] * inv)};
          w[g][0] = __builtin_bit_cast(unsigned, lo);
          w[g][1] = __builtin_bit_cast(unsigned, hi);
        }
        // after the swaps: half 0 holds d-groups 0 and 1 complete (its own first halves + half 1's second halves),
        // half 1 holds d-groups 2 and 3
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const auto sw = __builtin_amdgcn_permlane32_swap(w[g][k], w[g + 2][k], false, false);
            w[g][k] = sw[0];
            w[g + 2][k] = sw[1];
          }
        if (q0 + ql < N) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            // half 0: w[g] = own run (d 8g..8g+3), w[g+2] = half 1's run (d 8g+4..8g+7); half 1: w[g] = half 0's run of
            // group g+2 (d 8(g+2)..+3), w[g+2] = own (d 8(g+2)+4..+7)
            const uint4 v = {w[g][0], w[g][1], w[g + 2][0], w[g + 2][1]};
            *reinterpret_cast<uint4*>(orow + dt * 32 + 8 * (g + 2 * h)) = v;
          }
        }
      }
      if (h == 0 && q0 + ql < N) lse[(size_t)bh * N + q0 + ql] = mx * scale + __logf(sum);
    }
    ATTN_STAMP(5);
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// Round 5: the persistent forward, SPECIALISED on the number of key tiles and written as an explicit schedule.
// The kernel above serves any 128 < N <= 224 with one body: every tile is conditional (`kt < nkt`), so the compiler keeps the
// accumulators' zero fill (112 v_mov per head and wave), selects every score against "is this the last tile" (112 v_cndmask), and
// emits the three phases -- 28 MFMAs, then 112 maxima, then exponentials + 28 MFMAs -- one after the other: 779 vector
// instructions per head and wave where ~470 are needed, the matrix pipe busy 13 % of the time (profiles/r04_pmc_attn.txt).
// Here  * NKT is a template parameter: no per-tile conditions, the first MFMA of a chain takes a literal zero accumulator;
//       * the padded keys of the last tile are masked through that tile's accumulator SEED (-1e30 in their rows: they are
//         duplicates of key N - 1, finite, so their exponential is exactly 0): no select anywhere;
//       * the row maximum of tile kt - 1 is taken between the MFMAs of tile kt, and in the P V loop the exponentials of tile
//         kt + 1 are issued between the MFMAs of tile kt (an in-order wave that issues four MFMAs back to back waits 3 x 24
//         cycles for the pipe; two such waves per SIMD wait for each other too).  Every "MFMA | vector chunk" boundary is a
//         scheduling barrier: the order below is the order in the binary;
//       * K fragments are requested two tiles ahead.
// Same operand layouts, rounding points and LDS images as attn_fwd_persist_kernel (its results are bitwise equal where the
// summation order of the row sum allows: four partial sums here).
// ------------------------------------------------------------------------------------------
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int NKT>
__global__ __launch_bounds__(PF_WAVES * 64, 1) void attn_fwd_p2_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                       float* __restrict__ lse, int N, int H, int BH, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPAD = 224, IMG = NPAD * 128;       // one K or V image
  char* Qs = smem + 4 * IMG;                         // wave-private Q images behind the two (K, V) pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ld = 3 * H * HD;
  const int ql = lane & 31, h = lane >> 5;
  const int q0 = wave * 32;
  const float c2 = scale * 1.4426950408889634f;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int last_keys = N - (NKT - 1) * 32;
  char* myQ = Qs + wave * 4096;
  // accumulator seed of the LAST key tile: 0 in the rows of real keys, -1e30 in the padded ones
  f32x16 seed_last;
#pragma unroll
  for (int r = 0; r < 16; ++r) seed_last[r] = crow(r, h) < last_keys ? 0.f : -1.0e30f;

  unsigned kvoff[8], kvdst[8], qoff[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int q = wave + t * PF_WAVES;              // 0..55: < 28 -> K, else V
    const bool isv = q >= 28;
    const int piece = isv ? q - 28 : q;
    const int row = piece * 8 + (lane >> 3);
    const int rr = row < N ? row : N - 1;
    const int c = (lane & 7) ^ swzk(row);
    kvoff[t] = (unsigned)rr * (unsigned)(ld * 2) + (unsigned)(c * 16) + (unsigned)((isv ? 2 : 1) * H * HD * 2);
    kvdst[t] = (unsigned)((isv ? IMG : 0) + piece * 1024);   // wave-uniform
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = t * 8 + (lane >> 3);
    int gr = q0 + row;
    gr = gr < N ? gr : N - 1;
    const int c = (lane & 7) ^ swzk(row);
    qoff[t] = (unsigned)gr * (unsigned)(ld * 2) + (unsigned)(c * 16);
  }
  auto lds_of = [](const char* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)p); };   // LDS byte address
  auto head_base = [&](int bh) {
    const int b = bh / H, hd = bh - b * H;
    return reinterpret_cast<const char*>(qkv + (size_t)b * N * ld + hd * HD);   // wave-uniform
  };
  auto stage_q = [&](int bh) {
    const char* base = head_base(bh);
#pragma unroll
    for (int t = 0; t < 4; ++t) glds16_hidden(base, qoff[t], lds_of(myQ) + t * 1024);
  };

  int bh = blockIdx.x;
  if (bh >= BH) return;
  {
    const char* base = head_base(bh);
#pragma unroll
    for (int t = 0; t < 8; ++t) glds16_hidden(base, kvoff[t], lds_of(smem) + __builtin_amdgcn_readfirstlane(kvdst[t]));
  }
  stage_q(bh);
  int cur = 0;
  // The epilogue of a head -- normalise, convert, swap halves, store -- is DEFERRED into the S^T loop of the next head, a quarter
  // behind each of its first MFMA groups: that loop is bound by the matrix pipe (two waves x 28 MFMAs per SIMD) with the vector
  // unit idle, while at the end of a head the same 60 instructions and five stores ran with nothing beside them (0.6 us per head,
  // profiles/r05_c_attention_ab_and_stamps.txt).  Pending state: the output accumulators, 1 / sum, the log-sum-exp, two pointers.
  f32x16 po[2];
  float pinv = 0.f, plse = 0.f;
  bf16* prow = out;
  float* plsep = lse;
  bool pvalid = false;   // (lane-wise: false before the first head and in the lanes of padded queries)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) po[dt][r] = 0.f;
  unsigned pw[4][2];
  auto epi_convert = [&](const int dt) {   // run g = r >> 2 of this lane: d = 32 dt + 8 g + 4 h .. + 3, as two packed dwords
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pw[g][0] = cvt_pk_dword(po[dt][4 * g] * pinv, po[dt][4 * g + 1] * pinv);
      pw[g][1] = cvt_pk_dword(po[dt][4 * g + 2] * pinv, po[dt][4 * g + 3] * pinv);
    }
  };
  auto epi_store = [&](const int dt) {     // lane halves swap runs: two runs of EIGHT consecutive d per lane, 16-byte stores
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const auto sw = __builtin_amdgcn_permlane32_swap(pw[g][k2], pw[g + 2][k2], false, false);
        pw[g][k2] = sw[0];
        pw[g + 2][k2] = sw[1];
      }
    if (pvalid) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint4 v = {pw[g][0], pw[g][1], pw[g + 2][0], pw[g + 2][1]};
        *reinterpret_cast<uint4*>(prow + dt * 32 + 8 * (g + 2 * h)) = v;
      }
      if (dt == 1 && h == 0) *plsep = plse;
    }
  };
#ifdef CARA_ATTN_STAMPS
  int slot = -1;
#endif
#ifdef CARA_ATTN_PRIO   // (A/B build: static priority for the second-dispatched waves)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  for (; bh < BH; bh += gridDim.x) {
    const int nxt = bh + gridDim.x;
    const bool has_nxt = nxt < BH;
#ifdef CARA_ATTN_STAMPS
    ++slot;
#endif
    ATTN_STAMP(0);
    // ONE barrier per head (see attn_fwd_persist_kernel): this head's K, V, Q have landed, the previous head is done with
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SB();
    ATTN_STAMP(1);
    const char* Ks = smem + cur * 2 * IMG;
    const char* Vs = Ks + IMG;
    const int b = bh / H, head = bh - b * H;
    const char* nbase = head_base(has_nxt ? nxt : bh);
    const unsigned nimg = lds_of(smem + (cur ^ 1) * 2 * IMG);
    // piece t (0..7) of the next head's K / V images.  UNCONDITIONAL: behind the last head the pieces re-read this head (the
    // other image pair is free then, nobody reads it) -- a branch here splits the head's body into basic blocks, and the
    // compiler then sinks the exponentials of a tile out from between the MFMAs into the block that uses them (seen in the
    // first build of this kernel: every second tile's MFMAs came back to back).  The kernel drains its queue before it ends.
    auto next_piece = [&](const int t) { glds16_hidden(nbase, kvoff[t], nimg + __builtin_amdgcn_readfirstlane(kvdst[t])); };
    // piece t (0..3) of this wave's Q rows of the next head.  The wave's Q image is free as soon as its fragments are in registers:
    // every caller sits behind an MFMA that consumed them (the compiler's wait for qf has passed by then)
    auto next_q = [&](const int t) { glds16_hidden(nbase, qoff[t], lds_of(myQ) + t * 1024); };

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(myQ + ro.o[ks]);
    // ---------------- S^T = K Q^T, with the running maximum of the previous tile between the MFMAs ----------------
    f32x16 s[NKT];
    bf16x8 ka[3][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[0][ks] = *reinterpret_cast<const bf16x8*>(Ks + ro.o[ks]);
    if (NKT > 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ka[1][ks] = *reinterpret_cast<const bf16x8*>(Ks + 4096 + ro.o[ks]);
    }
    float m0 = -3.0e38f, m1 = -3.0e38f;   // two chains of v_max3
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + 2 < NKT) {
        const char* kb_ = Ks + (kt + 2) * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ka[(kt + 2) % 3][ks] = *reinterpret_cast<const bf16x8*>(kb_ + ro.o[ks]);
      }
      if ((kt & 1) == 0) next_piece(kt >> 1);   // K / V pieces 0-3 behind the even tiles here, 4-7 in the P V loop
      else if (kt < 6) next_q(kt >> 1);          // the wave's Q rows of the next head: pieces 0-2 here, 3 in the P V loop
      SB();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[kt % 3][ks], qf[ks], ks == 0 ? (kt == NKT - 1 ? seed_last : zero) : s[kt], 0, 0, 0);
        SB();
        if (kt > 0) {   // a quarter of the previous tile's maximum
          m0 = fmaxf(fmaxf(m0, s[kt - 1][4 * ks]), s[kt - 1][4 * ks + 1]);
          m1 = fmaxf(fmaxf(m1, s[kt - 1][4 * ks + 2]), s[kt - 1][4 * ks + 3]);
          SB();
        }
      }
      // the previous head's epilogue, one quarter per tile
      if (kt == 0) epi_convert(0);
      if (kt == 1) epi_store(0);
      if (kt == 2) epi_convert(1);
      if (kt == 3) epi_store(1);
      if (kt < 4) SB();
    }
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      m0 = fmaxf(fmaxf(m0, s[NKT - 1][r]), s[NKT - 1][r + 1]);
      m1 = fmaxf(fmaxf(m1, s[NKT - 1][r + 2]), s[NKT - 1][r + 3]);
    }
    ATTN_STAMP(2);
    float mx = fmaxf(m0, m1);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxc = mx * c2;
    ATTN_STAMP(3);
    // ---------------- P V, the exponentials of tile kt + 1 between the MFMAs of tile kt ----------------
    float sum0 = 0.f, sum1 = 0.f, sum2 = 0.f, sum3 = 0.f;
    // (scalar on purpose: the packed form -- v_pk_fma_f32 / v_pk_add_f32 on register pairs, 12 % fewer vector instructions -- measured
    // SLOWER here, 20.4 -> 20.8 us alone and 16.9 -> 17.7 in the step: profiles/r05_n_packed_vector_math_in_attention.txt)
    auto exp4 = [&](f32x16& t, const int g) {   // registers 4g .. 4g+3 of a tile: p = exp2(s c2 - mx c2), into four partial sums
      t[4 * g] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[4 * g], c2, -mxc));
      t[4 * g + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[4 * g + 1], c2, -mxc));
      t[4 * g + 2] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[4 * g + 2], c2, -mxc));
      t[4 * g + 3] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[4 * g + 3], c2, -mxc));
      sum0 += t[4 * g];
      sum1 += t[4 * g + 1];
      sum2 += t[4 * g + 2];
      sum3 += t[4 * g + 3];
    };
    f32x16 o[2];
    bf16x8 pa[2], vf[2][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) exp4(s[0], g);
    pa[0] = pack8(s[0], 0);
    pa[1] = pack8(s[0], 1);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vf[st][dt] = tr_frag_at(Vs, to.lo[st][dt], to.hi[st][dt]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if ((kt & 1) == 0) next_piece(4 + (kt >> 1));
      else if (kt == 1) next_q(3);
      SB();
      // O^T = V^T P^T (d on the rows); the first tile's products start the chains from a literal zero
      bf16x8 vn[2][2];
      bf16x8 pn[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int st = i >> 1, dt = i & 1;
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][dt], pa[st], (kt == 0 && st == 0) ? zero : o[dt], 0, 0, 0);
        SB();
        if (kt + 1 < NKT) {
          // next tile: its V fragment i is requested, a quarter of its exponentials computed
          vn[st][dt] = tr_frag_at(Vs + (kt + 1) * 4096, to.lo[st][dt], to.hi[st][dt]);
          exp4(s[kt + 1], i);
          if (i == 1) pn[0] = pack8(s[kt + 1], 0);
          if (i == 3) pn[1] = pack8(s[kt + 1], 1);
          SB();
        }
      }
      if (kt + 1 < NKT) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          pa[st] = pn[st];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vf[st][dt] = vn[st][dt];
        }
      }
    }
    if (NKT < 6) next_q(2);                                      // (NKT = 5 has no tile 5 in its S^T loop)
#pragma unroll
    for (int t = 4 + (NKT + 1) / 2; t < 8; ++t) next_piece(t);   // (the pieces of tiles this NKT does not have)
    if (NKT < 7) {
#pragma unroll
      for (int t = (NKT + 1) / 2; t < 4; ++t) next_piece(t);
    }
    float sum = (sum0 + sum1) + (sum2 + sum3);
    sum += __shfl_xor(sum, 32, 64);
    ATTN_STAMP(4);
    // hand this head's epilogue to the next head's S^T loop (or to the tail behind the loop)
    po[0] = o[0];
    po[1] = o[1];
    pinv = 1.0f / sum;
    plse = mx * scale + __logf(sum);
    pvalid = q0 + ql < N;
    prow = out + (size_t)b * N * (H * HD) + head * HD + (size_t)(pvalid ? q0 + ql : N - 1) * (H * HD);
    plsep = lse + (size_t)bh * N + (pvalid ? q0 + ql : 0);
    ATTN_STAMP(5);
    cur ^= 1;
  }
  epi_convert(0);
  epi_store(0);
  epi_convert(1);
  epi_store(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the pieces issued behind the last head: no LDS-DMA may outlive the workgroup)
}

// ------------------------------------------------------------------------------------------
// backward, kernel 1: dK, dV.  One workgroup of 7 waves per (batch, head); wave w owns keys
// 32w..32w+31 and keeps dK^T, dV^T in accumulators while sweeping the query tiles.  Q and dO are
// staged in LDS once, row-major: plain ds_read_b128 rows feed S = Q K^T and dP = dO V^T, and the
// transposing read ds_read_b64_tr_b16 of the same images feeds dV^T += dO^T P and dK^T += Q^T dS
// (whose B operands are the P / dS accumulators).
// ------------------------------------------------------------------------------------------
constexpr int BWD_WAVES = 7;   // keys per workgroup = 7 * 32; blockIdx.y walks the key groups when N > 224
__host__ __device__ constexpr int dkv_lds_bytes(int npad) { return 2 * npad * 128 + 2 * npad * 4; }

__global__ __launch_bounds__(448, 2) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                              const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                              bf16* __restrict__ dqkv, int N, int H, float scale, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + npad * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * npad * 128);
  float* del_s = lse_s + npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
  const int ld = 3 * H * HD, ldo = H * HD;
  const bf16* qb = qkv + (size_t)b * N * ld + head * HD;
  const bf16* kb = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const bf16* ob = out + (size_t)b * N * ldo + head * HD;
  const bf16* dob = dout + (size_t)b * N * ldo + head * HD;

  stage_rows_swz(qb, ld, N, Qs, tid, 448, npad);
  stage_rows_swz(dob, ldo, N, dOs, tid, 448, npad);
  for (int row = tid; row < npad; row += 448) {
    const int n = row < N ? row : N - 1;
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(ob + (size_t)n * ldo + c * 8);
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dob + (size_t)n * ldo + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)a[j] * (float)g[j];
    }
    del_s[row] = dl;
    lse_s[row] = lse[(size_t)bh * N + n] * 1.4426950408889634f;
  }
  __syncthreads();

  const int kl = lane & 31, h = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  // r05: the workgroup walks the key groups of its head (gridDim.y = 1): Q / dO / delta are staged once per head, not once per group
  for (int kg = blockIdx.y;; kg += gridDim.y) {
  const int key0 = (kg * BWD_WAVES + wave) * 32;
  if (key0 >= N) break;
  const int key = key0 + kl;
  const int keyc = key < N ? key : N - 1;
  const bool kvalid = key < N;
  // B operands with the key on the lane: K[key][16ks + 8h + j], V likewise
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(kb + (size_t)keyc * ld + ks * 16 + h * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(vb + (size_t)keyc * ld + ks * 16 + h * 8);
  }
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[dt][r] = 0.f; dvt[dt][r] = 0.f; }

  const int nqt = npad >> 5;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int last_q = N - (nqt - 1) * 32;   // valid queries of the last tile (rows beyond are clamped duplicates)
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 32;
    const char* qblk = Qs + qt * 4096;
    const char* dblk = dOs + qt * 4096;
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qblk + ro.o[ks]);
      const bf16x8 da = *reinterpret_cast<const bf16x8*>(dblk + ro.o[ks]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], pacc, 0, 0, 0);
    }
    // layout: column (lane & 31) = key, row = query q0 + crow(r, h).  Rows q >= N carry clamped
    // duplicates of row N-1: force their P to zero (last query tile only; an invalid key zeroes the whole lane).
    f32x16 p, ds;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {   // registers 4*g4 .. +3 hold queries q0 + 8*g4 + 4h .. +3: one 16-byte LDS read each
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + q0 + 8 * g4 + 4 * h);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + q0 + 8 * g4 + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = 4 * g4 + k;
        float e = __builtin_amdgcn_exp2f(sacc[r] * c2 - l4[k]);
        if (qt == nqt - 1) e = crow(r, h) < last_q ? e : 0.f;
        e = kvalid ? e : 0.f;
        p[r] = e;
        ds[r] = e * (pacc[r] - d4[k]);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 pb = pack8(p, st), dsb = pack8(ds, st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 doa = tr_frag_at(dblk, to.lo[st][dt], to.hi[st][dt]);
        const bf16x8 qta = tr_frag_at(qblk, to.lo[st][dt], to.hi[st][dt]);
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa, pb, dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qta, dsb, dkt[dt], 0, 0, 0);
      }
    }
  }
  // dK[key][d] = scale * dK^T[d][key]; registers 4g..4g+3 hold d = dt*32 + 8g + 4h + (0..3)
  if (kvalid) {
    bf16* dk = dqkv + (size_t)(b * N + key) * ld + H * HD + head * HD;
    bf16* dv = dk + H * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * h;
        bf16x4 a = {(bf16)(dkt[dt][4 * g] * scale), (bf16)(dkt[dt][4 * g + 1] * scale),
                    (bf16)(dkt[dt][4 * g + 2] * scale), (bf16)(dkt[dt][4 * g + 3] * scale)};
        bf16x4 c = {(bf16)dvt[dt][4 * g], (bf16)dvt[dt][4 * g + 1], (bf16)dvt[dt][4 * g + 2], (bf16)dvt[dt][4 * g + 3]};
        *reinterpret_cast<bf16x4*>(dk + d) = a;
        *reinterpret_cast<bf16x4*>(dv + d) = c;
      }
  }
  }
}

// ------------------------------------------------------------------------------------------
// backward, kernel 2: dQ.  Same shape as the forward: a wave owns 32 queries; per key tile it
// recomputes S^T = K Q^T and dP^T = V dO^T (key on the accumulator row, query on the lane, so the
// row constants LSE and delta are lane-local scalars), forms dS^T in registers and feeds it as
// the A operand of dQ += dS K (B fragments from the transposed K image): no LDS round trip, no
// cross-wave sum, no atomics.
// ------------------------------------------------------------------------------------------

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                                 const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                 bf16* __restrict__ dqkv, int N, int H, float scale, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + npad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
  const int ld = 3 * H * HD, ldo = H * HD;
  const bf16* qb = qkv + (size_t)b * N * ld + head * HD;
  const bf16* kb = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const bf16* ob = out + (size_t)b * N * ldo + head * HD;
  const bf16* dob = dout + (size_t)b * N * ldo + head * HD;
  stage_rows_swz(kb, ld, N, Ks, tid, NW * 64, npad);
  stage_rows_swz(vb, ld, N, Vs, tid, NW * 64, npad);
  __syncthreads();

  const int ql = lane & 31, h = lane >> 5;
  // r05: the workgroup walks the query groups of its head (gridDim.y = 1): K / V are staged once per head
  for (int qg = blockIdx.y;; qg += gridDim.y) {
  const int q0 = (qg * NW + wave) * 32;
  if (q0 >= N) break;
  const int qrow = (q0 + ql) < N ? (q0 + ql) : N - 1;
  bf16x8 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qrow * ld + ks * 16 + h * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dob + (size_t)qrow * ldo + ks * 16 + h * 8);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(ob + (size_t)qrow * ldo + ks * 16 + h * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)dof[ks][j];
  }
  dl += __shfl_xor(dl, 32, 64);   // delta[q] = sum_d dO[q,d] O[q,d]
  const float lq = lse[(size_t)bh * N + qrow] * 1.4426950408889634f;
  const float c2 = scale * 1.4426950408889634f;

  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  const int nkt = npad >> 5;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int last_keys = N - (nkt - 1) * 32;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* kblk = Ks + kt * 4096;
    const char* vblk = Vs + kt * 4096;
    f32x16 sT, dpT;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kblk + ro.o[ks]);
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(vblk + ro.o[ks]);
      sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], sT, 0, 0, 0);
      dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dpT, 0, 0, 0);
    }
    // layout: column (lane & 31) = query, row = key kt*32 + crow(r, h); only the last tile has keys >= N
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = __builtin_amdgcn_exp2f(sT[r] * c2 - lq);
      if (kt == nkt - 1) e = crow(r, h) < last_keys ? e : 0.f;
      ds[r] = e * (dpT[r] - dl);
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 a = pack8(ds, st);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 kf = tr_frag_at(kblk, to.lo[st][dt], to.hi[st][dt]);
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, a, dq[dt], 0, 0, 0);   // dQ^T = K^T dS^T: d on the rows
      }
    }
  }
  // dQ^T layout: column (lane & 31) = QUERY, register r of lane half h = d = 32 dt + 8 (r >> 2) + 4 h + (r & 3).  As in the
  // forward: the lane halves swap runs of four d (v_permlane32_swap) and every lane stores two runs of eight consecutive
  // d per dt -- four 16-byte stores per lane instead of 32 two-byte ones.
  bf16* qrow_out = dqkv + (size_t)(b * N + qrow) * ld + head * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    unsigned w[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x2 lo = {(bf16)(dq[dt][4 * g] * scale), (bf16)(dq[dt][4 * g + 1] * scale)};
      const bf16x2 hi = {(bf16)(dq[dt][4 * g + 2] * scale), (bf16)(dq[dt][4 * g + 3] * scale)};
      w[g][0] = __builtin_bit_cast(unsigned, lo);
      w[g][1] = __builtin_bit_cast(unsigned, hi);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const auto sw = __builtin_amdgcn_permlane32_swap(w[g][k], w[g + 2][k], false, false);
        w[g][k] = sw[0];
        w[g + 2][k] = sw[1];
      }
    if (q0 + ql < N) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint4 v = {w[g][0], w[g][1], w[g + 2][0], w[g + 2][1]};
        *reinterpret_cast<uint4*>(qrow_out + dt * 32 + 8 * (g + 2 * h)) = v;
      }
    }
  }
  }
}

// ------------------------------------------------------------------------------------------
// Backward for 128 < N <= 224, the headline path: ONE persistent kernel instead of the pair above.  A 7-wave workgroup
// per CU walks the (batch, head) pairs (768 = 3 per CU); for each pair it runs the dK/dV sweep (wave w owns keys 32w..,
// phase A) and then the dQ sweep (wave w owns queries 32w.., phase B) on the SAME LDS images, so Q, K, V, dO, O are read
// from HBM once per head instead of twice (195 -> 116 MB per layer), and the next head's images arrive by hidden LDS-DMA
// while this head computes:
//   phase A reads Q, dO (images) + its keys' K, V rows (registers);   meanwhile K, V of THIS head land in their images
//   phase B reads K, V (images) + its queries' Q, dO rows (registers); meanwhile Q, dO, O of the NEXT head land
// delta = rowsum(dO . O) is formed from the O and dO images (2 threads per row) before phase A.  Every wait for a DMA
// piece is a vmcnt(0) placed just before a phase's stores are issued, when everything outstanding is old; four
// barriers per head.  LDS: five [224][64] images + lse + delta = 145 KiB.  The arithmetic of the two phases is that
// of attn_bwd_dkv_kernel / attn_bwd_dq_kernel<7> (same operand layouts and rounding points; delta is summed in another
// order, so results agree to fp32 rounding of that one scalar per row, not bitwise).  Deterministic: no atomics.
// ------------------------------------------------------------------------------------------
// 4 bytes per lane (256 B per wave instruction) by hidden LDS-DMA: the forward's LSE row of a head (any 4-byte alignment)
__device__ __forceinline__ void glds4_hidden(const char* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

template <bool TWO_BARRIERS>
__global__ __launch_bounds__(448, 1) void attn_bwd_fused_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                                const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16* __restrict__ dqkv, int N, int H, int BH, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPAD = 224, IMG = NPAD * 128;
  char* Qs = smem;
  char* dOs = smem + IMG;
  char* Os = smem + 2 * IMG;
  char* Ks = smem + 3 * IMG;
  char* Vs = smem + 4 * IMG;
  float* lse_s = reinterpret_cast<float*>(smem + 5 * IMG);
  float* del_s = lse_s + 256;   // (lse_s takes four 64-float DMA pieces)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ld = 3 * H * HD, ldo = H * HD;
  const int l31 = lane & 31, h = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const float nrscale = -1.f / scale;
  const RowOfs ro = row_ofs(lane);
  const TrOfs to = tr_ofs(lane);
  const int nt = (N + 31) >> 5;                 // query / key tiles (<= 7)
  const int last = N - (nt - 1) * 32;           // valid rows of the last tile
  const int t0 = wave * 32;                     // first key (phase A) / query (phase B) of this wave
  const int trow = t0 + l31 < N ? t0 + l31 : N - 1;
  const bool tvalid = t0 + l31 < N;

  // per-lane 32-bit byte offsets of this wave's DMA pieces (head-independent; the head's base pointers are wave-uniform).
  // An image = 28 pieces of 8 rows; wave w takes pieces w, w + 7, w + 14, w + 21 of every image.
  unsigned off_qkv[4], off_o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = (wave + t * 7) * 8 + (lane >> 3);
    const int rr = row < N ? row : N - 1;
    const int c = (lane & 7) ^ swzk(row);
    off_qkv[t] = (unsigned)rr * (unsigned)(ld * 2) + (unsigned)(c * 16);
    off_o[t] = (unsigned)rr * (unsigned)(ldo * 2) + (unsigned)(c * 16);
  }
  auto lds_of = [](const char* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)p); };
  auto dma_image = [&](const char* base, const unsigned (&off)[4], char* img) {
#pragma unroll
    for (int t = 0; t < 4; ++t) glds16_hidden(base, off[t], lds_of(img) + (unsigned)((wave + t * 7) * 1024));
  };
  auto qkv_base = [&](int bh) { const int b = bh / H, hd = bh - b * H; return reinterpret_cast<const char*>(qkv + (size_t)b * N * ld + hd * HD); };
  auto o_base = [&](const bf16* p_, int bh) { const int b = bh / H, hd = bh - b * H; return reinterpret_cast<const char*>(p_ + (size_t)b * N * ldo + hd * HD); };

  int bh = blockIdx.x;
  if (bh >= BH) return;
  // ---- prologue: what a "phase B" leaves behind for the next head ----
  bf16x8 kf[4], vf[4];
  // the head's LSE row: wave 0, four pieces of 64 floats straight into lse_s (raw; scaled to log2 units in the delta step)
  unsigned off_lse[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = t * 64 + lane;
    off_lse[t] = (unsigned)((i < N ? i : N - 1) * 4);
  }
  auto dma_lse = [&](int bh_) {
    const char* base = reinterpret_cast<const char*>(lse + (size_t)bh_ * N);
    if constexpr (TWO_BARRIERS) {
      // piece w by wave w (w < 4): the wave that scales entries 64 w .. 64 w + 63 in the delta step is the wave whose own wait covers them
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (wave == t && t * 64 < NPAD) glds4_hidden(base, off_lse[t], lds_of(reinterpret_cast<const char*>(lse_s)) + (unsigned)(t * 256));
    } else if (wave == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * 64 < NPAD) glds4_hidden(base, off_lse[t], lds_of(reinterpret_cast<const char*>(lse_s)) + (unsigned)(t * 256));
    }
  };
  {
    const char* qb = qkv_base(bh);
    dma_image(qb, off_qkv, Qs);
    dma_image(o_base(dout, bh), off_o, dOs);
    dma_image(o_base(out, bh), off_o, Os);
    const bf16* kb = reinterpret_cast<const bf16*>(qb) + H * HD;
    const bf16* vb = kb + H * HD;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kb + (size_t)trow * ld + ks * 16 + h * 8);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vb + (size_t)trow * ld + ks * 16 + h * 8);
    }
    dma_lse(bh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#ifdef CARA_ATTN_STAMPS
  int slot = -1;
  // (head slot 3 is never walked at 3 heads per workgroup: its first entries take the shader-clock counter at both ends of the
  // head loop, next to the 100-MHz one, so that the script can print the clock the kernel ran at)
  if (g_attn_stamp_buf && tid == 0) {
    g_attn_stamp_buf[((size_t)blockIdx.x * 4 + 3) * 8 + 0] = __builtin_amdgcn_s_memtime();
    g_attn_stamp_buf[((size_t)blockIdx.x * 4 + 3) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#ifdef CARA_ATTN_PRIO   // (A/B build: static priority for the second-dispatched waves, cdna_hip_programming.md T5 static form)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  for (; bh < BH; bh += gridDim.x) {
    const int nxt = bh + gridDim.x;
    const int b = bh / H, head = bh - b * H;
#ifdef CARA_ATTN_STAMPS
    ++slot;
#endif
    ATTN_STAMP(0);
    // T0: every wave is through with phase B of the previous head (K, V images free) and has seen its own pieces of this
    // head's Q, dO, O images land
    // TWO_BARRIERS (r05): no barrier here.  Each wave forms delta for the rows of ITS OWN DMA pieces (pieces w, w + 7, w + 14, w + 21 of
    // the O / dO images: 32 rows, two lanes per row) and scales the 64 entries of its own LSE piece: everything it reads has landed
    // behind its own wait at the end of the previous head's dQ sweep; nobody still reads lse_s / del_s of the previous head (their
    // last reads sit in front of its T3).  T1 below then publishes delta / lse AND stands for "every wave's pieces have landed, every
    // wave is through with the previous head".
    if constexpr (TWO_BARRIERS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // delta[row] = sum_d dO[row][d] O[row][d] from the images, two threads per row (4 chunks of 8 each); lse in log2 units
    {
      const int row = TWO_BARRIERS ? (wave + 7 * (lane >> 4)) * 8 + ((lane >> 1) & 7) : tid >> 1, half = tid & 1;
      float dl = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int off = swz128(row, half * 4 + c);
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Os + off);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dOs + off);
#pragma unroll
        for (int j = 0; j < 8; j += 2) dl = dot2_acc(bf16x2{a[j], a[j + 1]}, bf16x2{g[j], g[j + 1]}, dl);   // (one instruction per pair, not six)
      }
      dl += __shfl_xor(dl, 1, 64);
      // Row constants in the form the S and dP accumulators are SEEDED with (phase A reads them straight into the
      // accumulator registers): S' = Q K^T - lse / scale, so p = exp2(c2 S') needs no subtraction, and dP' = dO V^T - delta,
      // so dS = p dP'.  A padded query row (>= N) gets -1e30: its p is exactly 0, no select per element anywhere.
      if (half == 0) del_s[row] = -dl;
      if (tid < NPAD) lse_s[tid] = tid < N ? lse_s[tid] * nrscale : -1e30f;   // (each element touched by exactly one thread)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // T1
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(1);

    // ================= phase A: dK, dV of keys t0 .. t0 + 31 =================
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkt[dt][r] = 0.f; dvt[dt][r] = 0.f; }
    // (fully unrolled over the <= 7 tiles: the compiler then requests a tile's LDS fragments under the previous tile's
    // arithmetic -- 68 -> 65 us per launch)
#pragma unroll
    for (int qt = 0; qt < 7; ++qt) {
      if (qt >= nt) break;
      const int q0 = qt * 32;
      const char* qblk = Qs + qt * 4096;
      const char* dblk = dOs + qt * 4096;
      // accumulators seeded with the row constants: register 4 g4 + k of lane half h is query row q0 + 8 g4 + 4 h + k
      f32x16 sacc, pacc;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + q0 + 8 * g4 + 4 * h);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + q0 + 8 * g4 + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sacc[4 * g4 + k] = l4[k]; pacc[4 * g4 + k] = d4[k]; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qblk + ro.o[ks]);
        const bf16x8 da = *reinterpret_cast<const bf16x8*>(dblk + ro.o[ks]);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], pacc, 0, 0, 0);
      }
      if (qt == 0) {
        // K, V of THIS head into their images (needed by phase B): issued behind the LAST first-use of kf / vf (scheduling
        // fence), so that none of the waits hipcc puts in front of those uses covers these pieces
        __builtin_amdgcn_sched_barrier(0);
        const char* qb = qkv_base(bh);
        dma_image(qb + H * HD * 2, off_qkv, Ks);
        dma_image(qb + 2 * H * HD * 2, off_qkv, Vs);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (a lane whose key is padding -- t0 + l31 >= N, its K / V rows duplicates of row N - 1 -- carries finite values
      // that only ever reach its OWN columns of dK^T / dV^T, which are not stored; a padded QUERY row has p = 0 through its
      // seed: no select per element)
      // (scalar on purpose, and the file is built with -fno-slp-vectorize: a packed fp32 instruction beside MFMAs costs ~22 cycles more
      // than the two scalar ones it replaces -- MI355X_MICROARCH.md, price of one filler beside MFMAs; r05 measured both forms)
      unsigned pw[8], dw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(sacc[2 * i] * c2), e1 = __builtin_amdgcn_exp2f(sacc[2 * i + 1] * c2);
        pw[i] = pk16(f32x2{e0, e1});
        dw[i] = pk16(f32x2{e0 * pacc[2 * i], e1 * pacc[2 * i + 1]});
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 pb = dwords8(pw, st), dsb = dwords8(dw, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 doa = tr_frag_at(dblk, to.lo[st][dt], to.hi[st][dt]);
          const bf16x8 qta = tr_frag_at(qblk, to.lo[st][dt], to.hi[st][dt]);
          dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa, pb, dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qta, dsb, dkt[dt], 0, 0, 0);
        }
      }
    }
    ATTN_STAMP(2);
    // T2: this wave's pieces of K, V have landed (they are old by now); then every wave's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // TWO_BARRIERS: no barrier here either -- the staging below is this wave's own 4 KiB of the O image (read by everybody in the
    // delta step only, a barrier ago), the row reads behind it are of the Q / dO images; T3 then also stands for "every wave's K / V
    // pieces have landed" (each wave passed the wait above first).  A wave that is through with its sweep stores while the others
    // still sweep instead of waiting for them twice.
    if constexpr (TWO_BARRIERS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(3);
    // dK, dV of this wave's 32 keys leave as WHOLE 128-byte rows: the accumulators hold them transposed (key on the lane, a
    // lane's 4 consecutive d per register group), so a direct store is 32 instructions that touch 32 rows each (2.7 us of a
    // head's 22: the stores are issue-bound per cache line touched).  Instead each matrix goes through this wave's 4 KiB of the
    // O image (free between T1 and T3; swizzled like every image here) and out as 4 stores of 8 full rows.
    {
      char* stg = Os + wave * 4096;
      const int srow = lane >> 3, schunk = lane & 7;
#pragma unroll
      for (int m = 0; m < 2; ++m) {   // 0: dK (scaled), 1: dV
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16& acc = m == 0 ? dkt[dt] : dvt[dt];
            const float f = m == 0 ? scale : 1.f;
            const bf16x4 a = {(bf16)(acc[4 * g] * f), (bf16)(acc[4 * g + 1] * f), (bf16)(acc[4 * g + 2] * f), (bf16)(acc[4 * g + 3] * f)};
            *reinterpret_cast<bf16x4*>(stg + swz128(l31, dt * 4 + g) + h * 8) = a;     // d = dt 32 + 8 g + 4 h ..+3
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16* dst = dqkv + (size_t)(b * N + t0) * ld + (1 + m) * H * HD + head * HD + schunk * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint4 v = *reinterpret_cast<const uint4*>(stg + swz128(srow + 8 * i, schunk));
          if (t0 + srow + 8 * i < N) *reinterpret_cast<uint4*>(dst + (size_t)(srow + 8 * i) * ld) = v;
        }
        asm volatile("" ::: "memory");   // (the image is rewritten by the next matrix / the next head's DMA)
      }
    }

    // ================= phase B: dQ of queries t0 .. t0 + 31 =================
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(Qs + wave * 4096 + ro.o[ks]);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dOs + wave * 4096 + ro.o[ks]);
    }
    const float lq = lse_s[t0 + l31] * c2, dl = del_s[t0 + l31];   // (-lse in log2 units | -delta; a padded query: -inf-like)
    // seed of S^T in the LAST key tile: its padded key rows (duplicates of key N - 1) start at -1e30, so their p is exactly 0
    f32x16 seed_last;
#pragma unroll
    for (int r = 0; r < 16; ++r) seed_last[r] = crow(r, h) < last ? 0.f : -1e30f;
    ATTN_STAMP(4);
    // T3: every wave holds its Q / dO rows and row constants: the Q, dO, O images may take the next head
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(5);
    // The next head's images and key rows: 12 DMA pieces + 8 row loads + the LSE row per wave.  Issued in one go they hold the
    // wave's instruction stream for 3-4 us (the CU's load path takes them at ~70 GB/s and a wave issues in order: the dQ sweep
    // measured 9.0 us with them in front and 5.0 without); handed out over the key tiles -- slice j behind the MFMAs of tile j --
    // they go out under the arithmetic.
    const bool has_nxt = nxt < BH;
    const char* nqb = qkv_base(has_nxt ? nxt : bh);
    const char* ndob = o_base(dout, has_nxt ? nxt : bh);
    const char* nob = o_base(out, has_nxt ? nxt : bh);
    auto next_slice = [&](const int j) {   // j = 0 .. 5 (compile-time in the unrolled callers)
      if (!has_nxt) return;
      const int t = j & 3;
      if (j < 4) {
        glds16_hidden(nqb, off_qkv[t], lds_of(Qs) + (unsigned)((wave + t * 7) * 1024));
        glds16_hidden(ndob, off_o[t], lds_of(dOs) + (unsigned)((wave + t * 7) * 1024));
        glds16_hidden(nob, off_o[t], lds_of(Os) + (unsigned)((wave + t * 7) * 1024));
        const bf16* kb = reinterpret_cast<const bf16*>(nqb) + H * HD;
        kf[t] = *reinterpret_cast<const bf16x8*>(kb + (size_t)trow * ld + t * 16 + h * 8);
      } else if (j == 4) {
        const bf16* vb = reinterpret_cast<const bf16*>(nqb) + 2 * H * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(vb + (size_t)trow * ld + ks * 16 + h * 8);
      } else {
        dma_lse(nxt);
      }
    };
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    auto tile_b = [&](const int kt, const f32x16& seed) {
      const char* kblk = Ks + kt * 4096;
      const char* vblk = Vs + kt * 4096;
      f32x16 sT = seed, dpT;
#pragma unroll
      for (int r = 0; r < 16; ++r) dpT[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kblk + ro.o[ks]);
        const bf16x8 va = *reinterpret_cast<const bf16x8*>(vblk + ro.o[ks]);
        sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], sT, 0, 0, 0);
        dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dpT, 0, 0, 0);
      }
      unsigned dw[8];   // dS^T = exp2(c2 S^T + lq) (dP^T + dl): scalar fp32, converted in pairs
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[2 * i], c2, lq)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[2 * i + 1], c2, lq));
        dw[i] = pk16(f32x2{e0 * (dpT[2 * i] + dl), e1 * (dpT[2 * i + 1] + dl)});
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 a = dwords8(dw, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 kfr = tr_frag_at(kblk, to.lo[st][dt], to.hi[st][dt]);
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, a, dq[dt], 0, 0, 0);   // dQ^T = K^T dS^T
        }
      }
    };
    {
      f32x16 zero;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
      // whole key tiles (unrolled: a tile's LDS fragments are requested under the previous tile's arithmetic), then the last
      // one with its padded keys masked through the seed
#pragma unroll
      for (int kt = 0; kt < 6; ++kt) {
        if (kt >= nt - 1) break;
        next_slice(kt);
        tile_b(kt, zero);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)   // (the slices of tiles this N does not have)
        if (j >= nt - 1) next_slice(j);
      tile_b(nt - 1, seed_last);
    }
    ATTN_STAMP(6);
    // the next head's images and row fragments have landed (they are old by now): wait for them HERE, before this phase's
    // stores go out, so that no later wait ever has to cover a store
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(7);
    bf16* qrow_out = dqkv + (size_t)(b * N + trow) * ld + head * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      unsigned w[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x2 lo = {(bf16)(dq[dt][4 * g] * scale), (bf16)(dq[dt][4 * g + 1] * scale)};
        const bf16x2 hi = {(bf16)(dq[dt][4 * g + 2] * scale), (bf16)(dq[dt][4 * g + 3] * scale)};
        w[g][0] = __builtin_bit_cast(unsigned, lo);
        w[g][1] = __builtin_bit_cast(unsigned, hi);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto sw = __builtin_amdgcn_permlane32_swap(w[g][k], w[g + 2][k], false, false);
          w[g][k] = sw[0];
          w[g + 2][k] = sw[1];
        }
      if (tvalid) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint4 v = {w[g][0], w[g][1], w[g + 2][0], w[g + 2][1]};
          *reinterpret_cast<uint4*>(qrow_out + dt * 32 + 8 * (g + 2 * h)) = v;
        }
      }
    }
  }
#ifdef CARA_ATTN_STAMPS
  if (g_attn_stamp_buf && tid == 0) {
    g_attn_stamp_buf[((size_t)blockIdx.x * 4 + 3) * 8 + 2] = __builtin_amdgcn_s_memtime();
    g_attn_stamp_buf[((size_t)blockIdx.x * 4 + 3) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------
// (Round 5 also built this backward as FOUR waves of 64 keys / queries -- one wave per SIMD, two 32-row blocks per wave, every
// LDS fragment read feeding two MFMAs (-39 % LDS reads), the vector work of block 0 fenced between the MFMAs of block 1 -- and it
// LOST: 67-69 us per launch against 60-65, 55-58 against 48-52 in the step, three builds (profiles/
// r05_u_attention_bwd_four_waves_rejected.txt, with time stamps: dK/dV sweep 6.9 vs 6.2 us per head, dQ sweep 5.7-8.5 vs 4.7-7.0).
// Correct at the first run; 256 architectural VGPRs + 184 AGPRs with ~300 v_accvgpr moves per head.  A lone wave pays every LDS and
// transcendental latency itself, and 6 vector instructions with two v_exp per MFMA gap are twice what hides in a gap
// (MI355X_MICROARCH.md, one wave per SIMD: <= 5 fillers, one of them 8-cycle).  Removed; commit 22344d3 has the kernel.)
// ------------------------------------------------------------------------------------------
// (Round 5 built the same specialisation for this kernel -- NT as a template parameter, branch-free DMA issue, the S^T / dP^T
// MFMAs of key tile kt + 1 between the vector work of tile kt in the dQ sweep -- and it LOST: 66.7-68.8 us per launch against
// 62.9-63.5 for the body above, same box (profiles/r05_c_attention_ab_and_stamps.txt).  The cross-tile pipeline needs two more
// accumulator tiles; next to dK^T / dV^T (64 registers), the K / V rows (32) and the fragments in flight that is more than the 256
// registers of two waves per SIMD: the dK/dV sweep could not be pipelined at all (404 bytes of scratch), the dQ sweep only with
// this wave's K / V rows of the next head requested behind it (0.4 -> 0.95 us exposed per head), and the compiler's order for the
// unpipelined dK/dV sweep came out slower than the one it finds for the loop above (8.0 vs 5.7 us per head).  Removed.)
// ------------------------------------------------------------------------------------------
// The LAST block: only the cls row of its attention output can reach the logits (the proj / MLP half of that block
// already runs on the cls rows alone), i.e. ONE query per (batch, head) against all N keys.  The full kernels spend
// 20 + 53 us on 197 queries there; this pair is two streaming passes over K and V: a workgroup of four waves per
// (batch, head), lane = head dimension d, the waves share the key rows n = wave, wave + 4, ...; dot products over d are
// wave reductions.  Same rounding points as the MFMA kernels (P and dS go through bf16 before they multiply V / dO / K / q,
// the row sum is taken of the unrounded exponentials, outputs are bf16).
// Forward writes out[b * N + 0, head] and lse[b, head, 0] only; backward reads those rows only and writes ALL of this
// layer's dqkv (dQ of the other rows is zero: they had no query in play).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block4_reduce(float v, float* red, const bool is_max) {   // 256 threads, result to all
  v = is_max ? wave_max(v) : wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();   // (red may still be read from a previous reduction)
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// lane = (row of a group of eight, 16-byte chunk of the 128-byte head row): a wave instruction moves eight whole rows, a dot
// product over d is 8 multiply-adds per lane + a reduction over the 8 lanes of a row; all of a wave's loads of a sweep are
// requested before anything is reduced (as one row per wave instruction with 2-byte lanes the pair took 42 + 37 us: every
// row a dependent load -> reduce chain)
__device__ __forceinline__ float row8_sum(float v) {   // sum over the 8 lanes (chunks) of a row
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float rows_sum(float v) {   // sum over the 8 row slots of a wave (same chunk)
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                           const int N, const int H, const float scale) {
  __shared__ float sc[NMAX_LONG + 32];
  __shared__ float red[4];
  __shared__ float part[4][64];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r8 = lane >> 3, c = lane & 7;
  const int ld = 3 * H * HD;
  const bf16* base = qkv + (size_t)b * N * ld + h * HD + c * 8;
  const bf16* kb = base + H * HD;
  const bf16* vb = kb + H * HD;
  float q8[8];
  {
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(base);   // the cls row is row 0 of the sample
#pragma unroll
    for (int j = 0; j < 8; ++j) q8[j] = (float)qv[j] * scale;
  }
  const int nit = (N + 31) / 32;                                // this wave's row groups: rows (it * 4 + wave) * 8 + r8
  // in batches of eight row groups (all of N <= 256 in one): eight 16-byte loads per lane in flight, fragments in registers
  // (indexed by a loop that can end early they went to scratch memory)
  for (int it0 = 0; it0 < nit; it0 += 8) {
    bf16x8 kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = ((it0 + u) * 4 + wave) * 8 + r8;
      kv[u] = *reinterpret_cast<const bf16x8*>(kb + (size_t)(n < N ? n : N - 1) * ld);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += q8[j] * (float)kv[u][j];
      s = row8_sum(s);
      const int n = ((it0 + u) * 4 + wave) * 8 + r8;
      if (c == 0 && it0 + u < nit) sc[n] = n < N ? s : -3.0e38f;   // (padded rows: exp -> 0)
    }
  }
  __syncthreads();
  const int npad = nit * 32;
  float mx = -3.0e38f;
  for (int n = threadIdx.x; n < npad; n += 256) mx = fmaxf(mx, sc[n]);
  mx = block4_reduce(mx, red, true);
  float sum = 0.f;
  for (int n = threadIdx.x; n < npad; n += 256) {
    const float e = __expf(sc[n] - mx);
    sum += e;                                                   // (unrounded, as in the MFMA kernels)
    sc[n] = (float)(bf16)e;                                     // P as the bf16 operand of P V
  }
  sum = block4_reduce(sum, red, false);                         // (its barriers also publish the rounded P)
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int it0 = 0; it0 < nit; it0 += 8) {
    bf16x8 vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = ((it0 + u) * 4 + wave) * 8 + r8;
      vv[u] = *reinterpret_cast<const bf16x8*>(vb + (size_t)(n < N ? n : N - 1) * ld);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pn = it0 + u < nit ? sc[((it0 + u) * 4 + wave) * 8 + r8] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pn * (float)vv[u][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = rows_sum(acc[j]);
    if (r8 == 0) part[wave][c * 8 + j] = t;
  }
  __syncthreads();
  if (wave == 0) {
    const float o = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / sum;
    out[(size_t)b * N * (H * HD) + h * HD + lane] = (bf16)o;
    if (lane == 0) lse[(size_t)bh * N] = mx + __logf(sum);
  }
}

__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                           const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                           bf16* __restrict__ dqkv, const int N, const int H, const float scale) {
  __shared__ float part[4][64];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r8 = lane >> 3, c = lane & 7;
  const int ld = 3 * H * HD;
  const size_t row0 = (size_t)b * N;
  const bf16* base = qkv + row0 * ld + h * HD + c * 8;
  const bf16* kb = base + H * HD;
  const bf16* vb = kb + H * HD;
  bf16* dq = dqkv + row0 * ld + h * HD + c * 8;
  bf16* dk = dq + H * HD;
  bf16* dv = dk + H * HD;
  float qs8[8], do8[8];
  float delta;
  {
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(base);
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(out + row0 * (H * HD) + h * HD + c * 8);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dout + row0 * (H * HD) + h * HD + c * 8);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qs8[j] = (float)qv[j] * scale;
      do8[j] = (float)gv[j];
      d += do8[j] * (float)ov[j];
    }
    delta = row8_sum(d);                                        // (every row slot its own copy)
  }
  const float L = lse[(size_t)bh * N];
  const int nit = (N + 31) / 32;
  float dqa[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dqa[j] = 0.f;
  // in batches of four row groups: 8 loads of 16 bytes per lane in flight
  for (int it0 = 0; it0 < nit; it0 += 4) {
    bf16x8 kk[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = ((it0 + u) * 4 + wave) * 8 + r8;
      const size_t ro = (size_t)(n < N ? n : N - 1) * ld;
      kk[u] = *reinterpret_cast<const bf16x8*>(kb + ro);
      vv[u] = *reinterpret_cast<const bf16x8*>(vb + ro);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = ((it0 + u) * 4 + wave) * 8 + r8;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s += qs8[j] * (float)kk[u][j];
        dp += do8[j] * (float)vv[u][j];
      }
      s = row8_sum(s);
      dp = row8_sum(dp);
      const float p = __expf(s - L);
      const float pb = (float)(bf16)p;                          // P and dS as the bf16 operands they are in the MFMA kernels
      const float dsb = (float)(bf16)(p * (dp - delta));
      if (n < N && it0 + u < nit) {
        bf16x8 ov, kv, zv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ov[j] = (bf16)(pb * do8[j]);
          kv[j] = (bf16)(dsb * qs8[j]);
          zv[j] = (bf16)0.f;
          dqa[j] += dsb * (float)kk[u][j];
        }
        *reinterpret_cast<bf16x8*>(dv + (size_t)n * ld) = ov;
        *reinterpret_cast<bf16x8*>(dk + (size_t)n * ld) = kv;
        if (n > 0) *reinterpret_cast<bf16x8*>(dq + (size_t)n * ld) = zv;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = rows_sum(dqa[j]);
    if (r8 == 0) part[wave][c * 8 + j] = t;
  }
  __syncthreads();
  if (wave == 0)
    dqkv[row0 * ld + h * HD + lane] = (bf16)(((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * scale);
}

// diagnostic: stage a [N,64] matrix like the kernels do and return every lane's transposed fragment
__global__ void tr_frag_probe_kernel(const bf16* __restrict__ src, bf16* __restrict__ out, int N, int cbase, int base) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  stage_rows_swz(src, 64, N, smem, threadIdx.x, 64, NMAX);
  __syncthreads();
  const bf16x8 f = tr_frag_rm(smem, cbase, base, threadIdx.x & 63);
  for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = f[j];
}

}  // namespace

extern "C" int cara_debug_tr_frag(const void* src, void* out, int N, int cbase, int base, void* stream) {
  hipLaunchKernelGGL(tr_frag_probe_kernel, dim3(1), dim3(64), NMAX * 128, static_cast<hipStream_t>(stream), (const bf16*)src,
                     (bf16*)out, N, cbase, base);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

// CARA_ATTN_WAVES=4 selects the 4-wave workgroups (two per head at N = 197) for A/B measurements
static int attn_waves(int N) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("CARA_ATTN_WAVES");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 4 || forced == 7) return forced;
  return N > 128 ? 7 : 4;
}

constexpr int MAX_LDS = 160 * 1024;
static void attn_set_lds_limits() {
  static bool done = false;
  if (done) return;
  const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<4>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<7>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_long_kernel<7>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_persist_kernel), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_p2_kernel<5>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_p2_kernel<6>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_p2_kernel<7>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<false>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<true>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<4>), at, MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<7>), at, MAX_LDS);
  done = true;
}

extern "C" int cara_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream) {
  if (!qkv || !out || !lse || B <= 0 || H <= 0 || N <= 0 || N > NMAX_LONG) return CARA_E_ARG;
  attn_set_lds_limits();
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int npad = (N + 31) / 32 * 32, lds = 2 * npad * 128;
  // The two-sweep kernel is also the default for 128 < N <= 224: 123 VGPRs instead of 210 put two 7-wave
  // workgroups on a CU, which more than pays for computing Q K^T twice (same-box 36.9 vs 41.4 us at N = 197).
  // CARA_ATTN_LONG=0 keeps the register-resident score rows for A/B runs.  (Capping the backward kernels at
  // 128 VGPRs for the same reason: dK/dV spills 38 registers, 145 vs 101 us for the pair; dQ alone 43.7 vs
  // 44.6 us, i.e. nothing -- not done.)
  static int use_long = -1;
  if (use_long < 0) {
    const char* e = getenv("CARA_ATTN_LONG");
    use_long = e ? atoi(e) : 1;
  }
  static const int use_persist = [] { const char* e = getenv("CARA_ATTN_PERSIST"); return e ? atoi(e) : 1; }();
  if (use_persist && N > 128 && N <= NMAX) {
    // one workgroup per CU walks the (batch, head) pairs: K / V of the next pair stream in while this one computes
    const int BH = B * H, grid = BH < 256 ? BH : 256;
    // CARA_ATTN_FWD_V=1: the round-3 body (any tile count in one kernel) for A/B runs; default: the specialised schedule
    static const int fwd_v = [] { const char* e = getenv("CARA_ATTN_FWD_V"); return e ? atoi(e) : 2; }();
    const size_t plds = 4 * 224 * 128 + PF_WAVES * 4096;
    const int nkt = (N + 31) / 32;
    if (fwd_v == 1)
      hipLaunchKernelGGL(attn_fwd_persist_kernel, dim3(grid), dim3(PF_WAVES * 64), plds, st, (const bf16*)qkv, (bf16*)out, lse, N, H, BH, scale);
    else if (nkt == 7)
      hipLaunchKernelGGL(attn_fwd_p2_kernel<7>, dim3(grid), dim3(PF_WAVES * 64), plds, st, (const bf16*)qkv, (bf16*)out, lse, N, H, BH, scale);
    else if (nkt == 6)
      hipLaunchKernelGGL(attn_fwd_p2_kernel<6>, dim3(grid), dim3(PF_WAVES * 64), plds, st, (const bf16*)qkv, (bf16*)out, lse, N, H, BH, scale);
    else
      hipLaunchKernelGGL(attn_fwd_p2_kernel<5>, dim3(grid), dim3(PF_WAVES * 64), plds, st, (const bf16*)qkv, (bf16*)out, lse, N, H, BH, scale);
  } else if (N > NMAX || (use_long && N > 128))
    hipLaunchKernelGGL(attn_fwd_long_kernel<7>, dim3(B * H, 1), dim3(448), lds, st, (const bf16*)qkv, (bf16*)out, lse, N,
                       H, scale, npad);   // (the workgroup walks its head's query groups: K / V staged once per head)
  else if (attn_waves(N) == 7)
    hipLaunchKernelGGL(attn_fwd_kernel<7>, dim3(B * H, (N + 223) / 224), dim3(448), lds, st, (const bf16*)qkv, (bf16*)out, lse, N, H,
                       scale, npad);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(B * H, (N + 127) / 128), dim3(256), lds, st, (const bf16*)qkv, (bf16*)out, lse, N, H,
                       scale, npad);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                  int B, int N, int H, float scale, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || H <= 0 || N <= 0 || N > NMAX_LONG) return CARA_E_ARG;
  attn_set_lds_limits();
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int npad = (N + 31) / 32 * 32, lds = 2 * npad * 128;
  static const int use_fused = [] { const char* e = getenv("CARA_ATTN_PERSIST"); return e ? atoi(e) : 1; }();
  if (use_fused && N > 128 && N <= NMAX) {
    const int BH = B * H, grid = BH < 256 ? BH : 256;
    // CARA_ATTN_BWD_V=1: the four-barrier protocol of round 3 for A/B runs; default (r05): two barriers per head
    static const int bwd_v = [] { const char* e = getenv("CARA_ATTN_BWD_V"); return e ? atoi(e) : 2; }();
    if (bwd_v == 1)
      hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, dim3(grid), dim3(448), 5 * 224 * 128 + (256 + 224) * 4, st, (const bf16*)qkv, (const bf16*)out,
                         (const bf16*)dout, lse, (bf16*)dqkv, N, H, BH, scale);
    else
      hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, dim3(grid), dim3(448), 5 * 224 * 128 + (256 + 224) * 4, st, (const bf16*)qkv, (const bf16*)out,
                         (const bf16*)dout, lse, (bf16*)dqkv, N, H, BH, scale);
    CARA_CHECK_LAUNCH();
    return CARA_OK;
  }
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H, 1), dim3(448), dkv_lds_bytes(npad), st, (const bf16*)qkv,
                     (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, N, H, scale, npad);
  CARA_CHECK_LAUNCH();
  if (N > NMAX || attn_waves(N) == 7)
    hipLaunchKernelGGL(attn_bwd_dq_kernel<7>, dim3(B * H, 1), dim3(448), lds, st, (const bf16*)qkv,
                       (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, N, H, scale, npad);
  else
    hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, dim3(B * H, (N + 127) / 128), dim3(256), lds, st, (const bf16*)qkv,
                       (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, N, H, scale, npad);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}


extern "C" int cara_attention_cls_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream) {
  if (!qkv || !out || !lse || B <= 0 || H <= 0 || N <= 0 || N > NMAX_LONG) return CARA_E_ARG;
  hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3(B * H), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16*)qkv, (bf16*)out, lse, N, H,
                     scale);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_attention_cls_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                      int B, int N, int H, float scale, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || H <= 0 || N <= 0 || N > NMAX_LONG) return CARA_E_ARG;
  hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3(B * H), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16*)qkv, (const bf16*)out,
                     (const bf16*)dout, lse, (bf16*)dqkv, N, H, scale);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
