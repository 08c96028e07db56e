// Epilogue of the MFMA GEMMs (gfx950): one wave turns a ROWS x 64 fp32 sub-tile that it has just
// written to its private LDS image `stg` ([ROWS][64] floats) into the final global stores, with every
// global access (output, residual / pre-activation input, bias) 16 bytes wide on 128..256-byte
// contiguous row segments.
#pragma once
#include "common.h"

template <int EPI, int ROWS>
__device__ __forceinline__ void epilogue_rows(const cara_gemm_args& p, const float* stg, const int mbase,
                                              const int nbase, const int lane, const size_t coff = 0) {
  if constexpr (EPI == CARA_EPI_F32 || EPI == CARA_EPI_RESID) {
    // fp32 output: 4 rows x 256 B per pass, 16 B per lane
    const int c4 = (lane & 15) * 4, n = nbase + c4;
    const bool vec = (n + 4 <= p.N) && ((p.ldc & 3) == 0);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[k] = (n + k < p.N) ? p.bias[n + k] : 0.f;
    }
#pragma unroll 4
    for (int pass = 0; pass < ROWS / 4; ++pass) {
      const int row = pass * 4 + (lane >> 4), m = mbase + row;
      const f32x4 a = *reinterpret_cast<const f32x4*>(stg + row * 64 + c4);
      if (m >= p.M || n >= p.N) continue;
      float v[4] = {a[0] + bv[0], a[1] + bv[1], a[2] + bv[2], a[3] + bv[3]};
      const size_t o = (size_t)m * p.ldc + n;
      if constexpr (EPI == CARA_EPI_RESID) {
        const float rs = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.f;
        const float* xin = static_cast<const float*>(p.aux) + o;
        if (vec) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(xin);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = x[k] + rs * v[k];
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = (n + k < p.N) ? xin[k] + rs * v[k] : 0.f;
        }
      }
      float* dst = static_cast<float*>(p.C) + o + coff;
      if (vec) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (n + k < p.N) dst[k] = v[k];
      }
    }
  } else {
    // bf16 output(s): 8 rows x 128 B per pass, 16 B per lane
    const int c8 = (lane & 7) * 8, n = nbase + c8;
    const bool vec = (n + 8 <= p.N) && ((p.ldc & 7) == 0);
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = (p.bias && n + k < p.N) ? p.bias[n + k] : 0.f;
#pragma unroll 2
    for (int pass = 0; pass < ROWS / 8; ++pass) {
      const int row = pass * 8 + (lane >> 3), m = mbase + row;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + row * 64 + c8);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + row * 64 + c8 + 4);
      if (m >= p.M || n >= p.N) continue;
      float v[8] = {a0[0] + bv[0], a0[1] + bv[1], a0[2] + bv[2], a0[3] + bv[3],
                    a1[0] + bv[4], a1[1] + bv[5], a1[2] + bv[6], a1[3] + bv[7]};
      const size_t o = (size_t)m * p.ldc + n;
      bf16x8 out, out2;
      h16x8 outh;
      if constexpr (EPI == CARA_EPI_GELU_DG) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float g, gp;
          gelu_erf_both(v[k], g, gp);
          out[k] = (bf16)g;
          outh[k] = (h16)gp;
        }
      } else if constexpr (EPI == CARA_EPI_MULH) {
        const h16* gpp = static_cast<const h16*>(p.aux) + o;
        h16x8 gv;
        if (vec) {
          gv = *reinterpret_cast<const h16x8*>(gpp);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) gv[k] = (n + k < p.N) ? gpp[k] : (h16)0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = (bf16)(v[k] * (float)gv[k]);
      } else if constexpr (EPI == CARA_EPI_BF16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = (bf16)v[k];
      } else if constexpr (EPI == CARA_EPI_GELU) {
        // h = gelu of the UNROUNDED fp32 pre-activation (rounding u first costs 3.4e-3 of logit error at
        // depth 12, oracle study in DESIGN.md); the bf16 copy of u is kept for gelu'(u) in backward only
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          out2[k] = (bf16)v[k];
          out[k] = (bf16)gelu_erf(v[k]);
        }
      } else {  // CARA_EPI_DGELU
        const bf16* up = static_cast<const bf16*>(p.aux) + o;
        bf16x8 u;
        if (vec) {
          u = *reinterpret_cast<const bf16x8*>(up);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) u[k] = (n + k < p.N) ? up[k] : (bf16)0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = (bf16)(v[k] * gelu_erf_grad((float)u[k]));
      }
      // C as K-panel-major [N/32][c_panels][32] (the next GEMM's A operand): the 8 columns stay inside one panel
      bf16* dst = p.c_panels ? static_cast<bf16*>(p.C) + ((size_t)(n >> 5) * p.c_panels + m) * 32 + (n & 31)
                             : static_cast<bf16*>(p.C) + o + coff;
      if (vec) {
        *reinterpret_cast<bf16x8*>(dst) = out;
        if constexpr (EPI == CARA_EPI_GELU) {
          if (p.C2) *reinterpret_cast<bf16x8*>(static_cast<bf16*>(p.C2) + o) = out2;   // (NULL: inference, u is not kept)
        }
        if constexpr (EPI == CARA_EPI_GELU_DG) {
          if (p.C2) *reinterpret_cast<h16x8*>(static_cast<h16*>(p.C2) + o) = outh;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (n + k < p.N) {
            dst[k] = out[k];
            if constexpr (EPI == CARA_EPI_GELU) {
              if (p.C2) (static_cast<bf16*>(p.C2) + o)[k] = out2[k];
            }
            if constexpr (EPI == CARA_EPI_GELU_DG) {
              if (p.C2) (static_cast<h16*>(p.C2) + o)[k] = outh[k];
            }
          }
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void epilogue_64x64(const cara_gemm_args& p, const float* stg, const int mbase,
                                               const int nbase, const int lane) {
  epilogue_rows<EPI, 64>(p, stg, mbase, nbase, lane);
}


// ---------------------------------------------------------------------------------------------------------------
// Fast epilogue for the bf16 outputs of INTERIOR wave tiles (CARA_EPI_BF16, CARA_EPI_GELU): rocprofv3 counted ~700 VALU
// instructions per wave and tile in the generic path above (12 per output element: fp32 staging moves, per-row bounds
// tests, 64-bit address arithmetic), and every one of them is exposed -- the GELU arithmetic alone, 12 more per element,
// is 15 us of a 100 us fc1 product (tools: CARA_ABLATE_GELU).  Here bias, GELU and the bf16 conversion happen in the
// MFMA accumulator layout (lane = column, 4 consecutive rows per register quad), the bf16 VALUES go through a wave-
// private LDS image with 2-byte writes (rows padded to 144 B: the four row groups of a write land on disjoint banks),
// and come back as 16-byte row pieces for the global stores: ~2 VALU per element besides the GELU itself.
// acc: the wave's 64 x 64 sub-tile as [4][4] accumulators of v_mfma_f32_16x16x32 (row = 16 i + 4 (lane >> 4) + r,
// column = 16 j + (lane & 15)).  stg: >= 32 * 144 bytes of LDS owned by this wave.  Requires mbase + 64 <= M,
// nbase + 64 <= N, ldc % 8 == 0 (callers test; edge tiles take the generic path).
// ---------------------------------------------------------------------------------------------------------------
constexpr int EPI_FAST_ROW_BYTES = 144;
constexpr int EPI_FAST_WAVE_BYTES = 32 * EPI_FAST_ROW_BYTES;

template <int EPI>
__device__ __forceinline__ void epilogue_fast_bf16(const cara_gemm_args& p, const f32x4 (&acc)[4][4], char* stg, const int mbase,
                                                   const int nbase, const int lane, const size_t coff) {
  static_assert(EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU, "bf16 outputs computed from the accumulator alone");
  const int fr = lane & 15, fq = lane >> 4;
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[nbase + j * 16 + fr] : 0.f;
  // this lane's write position inside a 32-row image: row 4 fq + r of row tile i2, column 16 j + fr
  char* wbase = stg + (4 * fq) * EPI_FAST_ROW_BYTES + fr * 2;
  // and its read position: row (lane >> 3) of each 8-row pass, 16-byte chunk lane & 7
  const char* rbase = stg + (lane >> 3) * EPI_FAST_ROW_BYTES + (lane & 7) * 16;
  const int rrow = lane >> 3, rcol = (lane & 7) * 8;
  constexpr int NOUT = EPI == CARA_EPI_GELU ? 2 : 1;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      // o == 0: C (h = gelu(u) for CARA_EPI_GELU); o == 1: C2 = u, skipped when the pre-activation is not kept
      if (o == 1 && !p.C2) continue;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[half * 2 + i2][j][r] + bv[j];
            if (EPI == CARA_EPI_GELU && o == 0) v[r] = gelu_erf(v[r]);
          }
          char* w = wbase + (i2 * 16) * EPI_FAST_ROW_BYTES + j * 32;
          const unsigned p01 = cvt_pk_dword(v[0], v[1]), p23 = cvt_pk_dword(v[2], v[3]);
          *reinterpret_cast<unsigned short*>(w) = (unsigned short)p01;
          *reinterpret_cast<unsigned short*>(w + EPI_FAST_ROW_BYTES) = (unsigned short)(p01 >> 16);
          *reinterpret_cast<unsigned short*>(w + 2 * EPI_FAST_ROW_BYTES) = (unsigned short)p23;
          *reinterpret_cast<unsigned short*>(w + 3 * EPI_FAST_ROW_BYTES) = (unsigned short)(p23 >> 16);
        }
      // (wave-private image: the wave's own LDS operations complete in order, no barrier; the fences keep the COMPILER from
      // reordering the 2-byte stores and the 16-byte loads, different types to its alias analysis)
      asm volatile("" ::: "memory");
      bf16* out = static_cast<bf16*>(o == 0 ? p.C : p.C2);
      const bool panels = o == 0 && p.c_panels;   // C as K-panel-major [N/32][c_panels][32]; C2 / aux keep the row-major ldc
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const bf16x8 val = *reinterpret_cast<const bf16x8*>(rbase + pass * 8 * EPI_FAST_ROW_BYTES);
        const int m = mbase + half * 32 + pass * 8 + rrow, n = nbase + rcol;
        bf16* dst = panels ? out + ((size_t)(n >> 5) * p.c_panels + m) * 32 + (n & 31) : out + (size_t)m * p.ldc + n + (o == 0 ? coff : 0);
        *reinterpret_cast<bf16x8*>(dst) = val;
      }
      asm volatile("" ::: "memory");
    }
  }
}

// The same for NT row tiles of 16 rows, one row tile at a time through a 16-row image per output (the 160-row tile: 5 row tiles)
template <int EPI, int NT>
__device__ __forceinline__ void epilogue_fast_bf16_rt(const cara_gemm_args& p, const f32x4 (&acc)[NT][4], char* stg, const int mbase,
                                                      const int nbase, const int lane, const size_t coff) {
  static_assert(EPI == CARA_EPI_BF16 || EPI == CARA_EPI_GELU || EPI == CARA_EPI_GELU_DG, "bf16 outputs computed from the accumulator alone");
  const int fr = lane & 15, fq = lane >> 4;
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[nbase + j * 16 + fr] : 0.f;
  char* wbase = stg + (4 * fq) * EPI_FAST_ROW_BYTES + fr * 2;
  const char* rbase = stg + (lane >> 3) * EPI_FAST_ROW_BYTES + (lane & 7) * 16;
  const int rrow = lane >> 3, rcol = (lane & 7) * 8;
  if constexpr (EPI == CARA_EPI_GELU_DG) {
    // h = gelu(u) and gelu'(u) from ONE evaluation of the shared terms, both 16-bit images written in the same pass (the two
    // 16-row halves of the staging area); the derivative as IEEE half.  p.C2 == NULL (inference): h only
    const bool keep = p.C2 != nullptr;   // (wave-uniform)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g[4], gp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r] + bv[j];
          if (keep) gelu_erf_both(v, g[r], gp[r]);
          else g[r] = gelu_erf(v);
        }
        char* w = wbase + j * 32;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const unsigned pr = cvt_pk_dword(g[r], g[r + 1]);
          *reinterpret_cast<unsigned short*>(w + r * EPI_FAST_ROW_BYTES) = (unsigned short)pr;
          *reinterpret_cast<unsigned short*>(w + (r + 1) * EPI_FAST_ROW_BYTES) = (unsigned short)(pr >> 16);
        }
        if (keep) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<h16*>(w + (16 + r) * EPI_FAST_ROW_BYTES) = (h16)gp[r];
        }
      }
      asm volatile("" ::: "memory");
      bf16x8 v8[2][2];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) v8[o][pass] = *reinterpret_cast<const bf16x8*>(rbase + (o * 16 + pass * 8) * EPI_FAST_ROW_BYTES);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int m = mbase + i * 16 + pass * 8 + rrow, n = nbase + rcol;
        bf16* out = static_cast<bf16*>(p.C);
        bf16* dst = p.c_panels ? out + ((size_t)(n >> 5) * p.c_panels + m) * 32 + (n & 31) : out + (size_t)m * p.ldc + n + coff;
        *reinterpret_cast<bf16x8*>(dst) = v8[0][pass];
        if (keep) *reinterpret_cast<bf16x8*>(static_cast<bf16*>(p.C2) + (size_t)m * p.ldc + n) = v8[1][pass];   // (16 bytes of halves)
      }
    }
    return;
  }
  constexpr int NOUT = EPI == CARA_EPI_GELU ? 2 : 1;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      if (o == 1 && !p.C2) continue;
      char* w0 = wbase + (o * 16) * EPI_FAST_ROW_BYTES;   // the two outputs use the two 16-row halves of the image
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bv[j];
          if (EPI == CARA_EPI_GELU && o == 0) v[r] = gelu_erf(v[r]);
        }
        char* w = w0 + j * 32;
        const unsigned p01 = cvt_pk_dword(v[0], v[1]), p23 = cvt_pk_dword(v[2], v[3]);
        *reinterpret_cast<unsigned short*>(w) = (unsigned short)p01;
        *reinterpret_cast<unsigned short*>(w + EPI_FAST_ROW_BYTES) = (unsigned short)(p01 >> 16);
        *reinterpret_cast<unsigned short*>(w + 2 * EPI_FAST_ROW_BYTES) = (unsigned short)p23;
        *reinterpret_cast<unsigned short*>(w + 3 * EPI_FAST_ROW_BYTES) = (unsigned short)(p23 >> 16);
      }
      // (compiler fences: the 2-byte stores and the 16-byte loads of the image are different types to the alias analysis)
      asm volatile("" ::: "memory");
      bf16* out = static_cast<bf16*>(o == 0 ? p.C : p.C2);
      const bool panels = o == 0 && p.c_panels;
      bf16x8 v8[2];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) v8[pass] = *reinterpret_cast<const bf16x8*>(rbase + (o * 16 + pass * 8) * EPI_FAST_ROW_BYTES);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int m = mbase + i * 16 + pass * 8 + rrow, n = nbase + rcol;
        bf16* dst = panels ? out + ((size_t)(n >> 5) * p.c_panels + m) * 32 + (n & 31) : out + (size_t)m * p.ldc + n + (o == 0 ? coff : 0);
        *reinterpret_cast<bf16x8*>(dst) = v8[pass];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogues that READ a second operand (CARA_EPI_RESID: the fp32 residual stream, CARA_EPI_DGELU: the bf16 pre-activation)
// for INTERIOR wave tiles.  In the generic path above every group of passes waits for its own loads: in-kernel time
// stamps put a workgroup's residual epilogue at 12.8 us (25 us from cold caches), four exposed memory latencies in a row.
// Here the operand rows of a whole HALF-row pass group are requested at once, and the NEXT group's while the current one
// is converted and stored: one exposed latency per tile.  No bounds tests (interior), the per-sample DropPath scale
// without an integer division per row (a wave tile of at most 128 rows spans at most two samples when rows_per_sample
// >= its height; callers test that), row addresses by increments.
// acc: [NT][4] accumulators of v_mfma_f32_16x16x32 (row = 16 i + 4 (lane >> 4) + r, column = 16 j + (lane & 15));
// stg: a wave-private [GROUP * 16][64] fp32 image; GROUP = row tiles per pass group (NT % GROUP == 0).
// ---------------------------------------------------------------------------------------------------------------
template <int EPI, int NT, int GROUP>
__device__ __forceinline__ void epilogue_interior_aux(const cara_gemm_args& p, const f32x4 (&acc)[NT][4], float* stg, const int mbase,
                                                      const int nbase, const int lane) {
  static_assert(EPI == CARA_EPI_RESID || EPI == CARA_EPI_DGELU || EPI == CARA_EPI_MULH, "epilogues with an input operand");
  static_assert(NT % GROUP == 0, "whole pass groups");
  constexpr int ROWS = GROUP * 16, NG = NT / GROUP;
  const int fr = lane & 15, fq = lane >> 4;
  if constexpr (EPI == CARA_EPI_RESID) {
    constexpr int NP = ROWS / 4;   // passes of 4 rows x 256 B, 16 B per lane
    const int c4 = (lane & 15) * 4, rl = lane >> 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + nbase + c4);
    // DropPath scale of this lane's rows: sample s0 for rows below `edge`, s0 + 1 from there on
    float r0 = 1.f, r1 = 1.f;
    int edge = 0x7fffffff;
    if (p.rowscale) {
      const int s0 = mbase / p.rows_per_sample;   // wave-uniform: one scalar division per tile
      edge = (s0 + 1) * p.rows_per_sample;
      r0 = p.rowscale[s0];
      r1 = edge < mbase + NT * 16 ? p.rowscale[s0 + 1] : r0;
    }
    const float* xin = static_cast<const float*>(p.aux) + (size_t)(mbase + rl) * p.ldc + nbase + c4;
    float* dst = static_cast<float*>(p.C) + (size_t)(mbase + rl) * p.ldc + nbase + c4;
    const size_t step = (size_t)4 * p.ldc;
    f32x4 x[2][NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) x[0][q] = *reinterpret_cast<const f32x4*>(xin + q * step);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int i = 0; i < GROUP; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[g * GROUP + i][j][r];
      // (compiler fence: the scalar stores and the 16-byte loads of the image are different types to the alias analysis, which
      // once moved the first load above the last store)
      asm volatile("" ::: "memory");
      if (g + 1 < NG) {
#pragma unroll
        for (int q = 0; q < NP; ++q) x[(g + 1) & 1][q] = *reinterpret_cast<const f32x4*>(xin + ((g + 1) * NP + q) * step);
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(stg + (q * 4 + rl) * 64 + c4);
        const float rs = mbase + g * ROWS + q * 4 + rl < edge ? r0 : r1;
        const f32x4 xv = x[g & 1][q];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = xv[k] + rs * (a[k] + bv[k]);
        *reinterpret_cast<f32x4*>(dst + (g * NP + q) * step) = o;
      }
      asm volatile("" ::: "memory");
    }
  } else {
    constexpr int NP = ROWS / 8;   // passes of 8 rows x 128 B, 16 B per lane
    const int c8 = (lane & 7) * 8, rl = lane >> 3;
    const bf16* up = static_cast<const bf16*>(p.aux) + (size_t)(mbase + rl) * p.ldc + nbase + c8;
    const size_t step = (size_t)8 * p.ldc;
    bf16x8 u[2][NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) u[0][q] = *reinterpret_cast<const bf16x8*>(up + q * step);
    const int n = nbase + c8;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int i = 0; i < GROUP; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[g * GROUP + i][j][r];
      // (compiler fence: the scalar stores and the 16-byte loads of the image are different types to the alias analysis, which
      // once moved the first load above the last store)
      asm volatile("" ::: "memory");
      if (g + 1 < NG) {
#pragma unroll
        for (int q = 0; q < NP; ++q) u[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(up + ((g + 1) * NP + q) * step);
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + (q * 8 + rl) * 64 + c8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + (q * 8 + rl) * 64 + c8 + 4);
        const bf16x8 uv = u[g & 1][q];
        bf16x8 out;
        if constexpr (EPI == CARA_EPI_MULH) {   // (the saved derivative, IEEE half in the same 16 bytes)
          const h16x8 gv = __builtin_bit_cast(h16x8, uv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            out[k] = (bf16)(a0[k] * (float)gv[k]);
            out[4 + k] = (bf16)(a1[k] * (float)gv[4 + k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            out[k] = (bf16)(a0[k] * gelu_erf_grad((float)uv[k]));
            out[4 + k] = (bf16)(a1[k] * gelu_erf_grad((float)uv[4 + k]));
          }
        }
        const int m = mbase + g * ROWS + q * 8 + rl;
        bf16* dst = p.c_panels ? static_cast<bf16*>(p.C) + ((size_t)(n >> 5) * p.c_panels + m) * 32 + (n & 31)
                               : static_cast<bf16*>(p.C) + (size_t)m * p.ldc + n;
        *reinterpret_cast<bf16x8*>(dst) = out;
      }
      asm volatile("" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// CARA_EPI_MULH with EPILOGUE RIDERS (cara_gemm_args::er_*): the fc2 dX tile that turns its accumulators into dH = acc * gelu'(u)
// also reads the h = gelu(u) tile at the same coordinates and multiplies both -- transposed -- by its rows of T^T (fc1's) and G'^T
// (fc2's): partial dVs_fc1 = dH^T T and dU_fc2 = h^T G' for its 64 columns.  The 77 MB of dH and the 77 MB of h that those two
// products used to re-read per block (as workgroups riding in the fc1 dX launch, 4-KiB strided tiles) are read here once, as 16-byte
// row pieces next to the saved derivative, and never again.
// Per 16-row tile of the wave's NT: accumulators -> fp32 image -> 16-byte row pieces (as epilogue_interior_aux), dH / h as bf16 ->
// two 16 x 64 bf16 images (rows padded to 144 B) -> one transposing LDS read per 16-column tile and product = the A operand of a
// K = 16 MFMA (lane fr: column fr, rows 4 fq .. 4 fq + 3), B operand = 8 bytes of T^T / G'^T row fr.  The reads and MFMAs of a row
// tile are issued one tile LATER, between the next tile's image writes and its arithmetic: the LDS round trip hides under VALU work.
// Everything is wave-private (a wave's LDS operations complete in order; compiler fences between the differently typed accesses).
// INTERIOR = false: rows >= M contribute zeros and are not stored.  Callers guarantee N % 64 == 0 for the wave tile, ldc % 8 == 0,
// M % 4 == 0.  Results: rv[j] / ru[j] = the wave's partial [16 j + 4 fq + reg][fr] sums, cs[j] = this lane's share of the column sum
// of the bf16 dH (column nbase + 16 j + fr, the rows 4 fq .. 4 fq + 3 of every row tile).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ER_ROW_BYTES = 144;
constexpr int ER_IMG_BYTES = 16 * ER_ROW_BYTES;
constexpr int ER_WAVE_BYTES = 2 * ER_IMG_BYTES;   // 4608: the dH image and the h image
constexpr int ER_STG_BYTES = 4096 + 256;          // per wave: the 16 x 64 fp32 staging image + the column sums of the wave-row exchange
constexpr int ER_LDS_BYTES = 4 * (ER_STG_BYTES + ER_WAVE_BYTES);

template <int NT, bool INTERIOR>
__device__ __forceinline__ void epilogue_mulh_riders(const cara_gemm_args& p, const f32x4 (&acc)[NT][4], float* stg, char* img, const int mbase,
                                                     const int nbase, const int lane, f32x4 (&rv)[4], f32x4 (&ru)[4], float (&cs)[4]) {
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  const int fr = lane & 15, fq = lane >> 4;
  const int c8 = (lane & 7) * 8, rl = lane >> 3;
  const int n = nbase + c8;
  // Addresses: wave-uniform 64-bit bases (scalar registers) + ONE 32-bit byte offset per lane and layout.  Row-major [M, ldc] arrays
  // (the saved derivative; h and C unless K-panel-major) share the offset of (row rl of the wave tile, column n); T^T and G'^T share
  // (row fr, column 4 fq).  Every array spans < 4 GiB (callers check).
  const unsigned ldc2 = (unsigned)p.ldc * 2u;
  const unsigned off_rm = (unsigned)(mbase + rl) * ldc2 + (unsigned)n * 2u;
  const unsigned off_hp = p.er_h_panels ? ((unsigned)(n >> 5) * (unsigned)p.er_h_panels + (unsigned)(mbase + rl)) * 64u + (unsigned)(n & 31) * 2u : off_rm;
  const unsigned off_cp = p.c_panels ? ((unsigned)(n >> 5) * (unsigned)p.c_panels + (unsigned)(mbase + rl)) * 64u + (unsigned)(n & 31) * 2u : off_rm;
  const unsigned hstep = p.er_h_panels ? 64u : ldc2, cstep = p.c_panels ? 64u : ldc2;   // bytes per row
  const unsigned off_tg = (unsigned)fr * (unsigned)p.er_ldg * 2u + (unsigned)(mbase + 4 * fq) * 2u;
  const char* __restrict__ gp_b = static_cast<const char*>(p.aux);
  const char* __restrict__ h_b = static_cast<const char*>(p.er_h);
  char* __restrict__ c_b = static_cast<char*>(p.C);
  const char* __restrict__ tt_b = static_cast<const char*>(p.er_Tt);
  const char* __restrict__ gt_b = static_cast<const char*>(p.er_Gt);
  const bf16x4 z4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  h16x8 gv[2];
  bf16x8 hv[2];
  bf16x4 bt, bg;
  // rows of an edge tile: the wave-uniform count of valid rows below mbase (INTERIOR: all)
  const int rows_ok = p.M - mbase;
  auto request = [&](int i) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int r = i * 16 + q * 8;   // (row of the wave tile, without the lane's rl)
      if (!INTERIOR) r = r + rl < rows_ok ? r : rows_ok - 1 - rl;   // clamp to the last valid row: a readable address, masked below
      // (32-bit sums: a clamped row may lie ABOVE the wave tile's first row, r < 0, and the sum still lands inside the array)
      gv[q] = *reinterpret_cast<const h16x8*>(gp_b + (size_t)(unsigned)((unsigned)r * ldc2 + off_rm));
      hv[q] = *reinterpret_cast<const bf16x8*>(h_b + (size_t)(unsigned)((unsigned)r * hstep + off_hp));
    }
  };
  auto request_b = [&](int i) {
    const bool ok = INTERIOR || i * 16 + 4 * fq < rows_ok;   // (M % 4 == 0: the four rows are valid together)
    bt = ok ? *reinterpret_cast<const bf16x4*>(tt_b + (size_t)(unsigned)(i * 32 + off_tg)) : z4;
    bg = ok ? *reinterpret_cast<const bf16x4*>(gt_b + (size_t)(unsigned)(i * 32 + off_tg)) : z4;
  };
  // the transposing reads of this lane: row 4 fq + (fr >> 2) of the image, 8 bytes at column 16 j + 4 (fr & 3)
  const char* tbase = img + (4 * fq + (fr >> 2)) * ER_ROW_BYTES + (fr & 3) * 8;
  char* wbase = img + rl * ER_ROW_BYTES + c8 * 2;
  auto products = [&]() {   // the images of the previous row tile x bt / bg; the column sums of dH from the same fragments
#if !(defined(CARA_ER_ABLATE) && (CARA_ER_ABLATE & 1))   // timing diagnostic (tools/build_variant.sh): without the products
    s16x4_t f4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f4[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4_t*)(tbase + j * 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x4 d = __builtin_bit_cast(bf16x4, f4[j]);
      rv[j] = mfma_16x16x16(d, bt, rv[j]);
      cs[j] += ((float)d[0] + (float)d[1]) + ((float)d[2] + (float)d[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) f4[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4_t*)(tbase + ER_IMG_BYTES + j * 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) ru[j] = mfma_16x16x16(__builtin_bit_cast(bf16x4, f4[j]), bg, ru[j]);
#endif
  };
  request(0);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * 64 + j * 16 + fr] = acc[i][j][r];
    asm volatile("" ::: "memory");
    if (i > 0) products();
    request_b(i);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + (q * 8 + rl) * 64 + c8);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + (q * 8 + rl) * 64 + c8 + 4);
      const int r = i * 16 + q * 8;
      const bool valid = INTERIOR || r + rl < rows_ok;
      bf16x8 out, hh = hv[q];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float o = (k < 4 ? a0[k & 3] : a1[k & 3]) * (float)gv[q][k];
        if (!INTERIOR) {
          o = valid ? o : 0.f;
          hh[k] = valid ? hh[k] : (bf16)0.f;
        }
        out[k] = (bf16)o;
      }
      if (valid) *reinterpret_cast<bf16x8*>(c_b + (size_t)(unsigned)((unsigned)r * cstep + off_cp)) = out;
      *reinterpret_cast<bf16x8*>(wbase + q * 8 * ER_ROW_BYTES) = out;
      *reinterpret_cast<bf16x8*>(wbase + ER_IMG_BYTES + q * 8 * ER_ROW_BYTES) = hh;
    }
    if (i + 1 < NT) request(i + 1);
    asm volatile("" ::: "memory");
  }
  products();
}
