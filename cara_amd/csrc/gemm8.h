// The MT x 256 x 64 one-workgroup-per-CU GEMM tile (gemm8.hip) as seen by the dispatcher in gemm.hip.
#pragma once
#include "common.h"

// one transposed skinny product riding in the launch (the fields of TsProblem, tskinny_body.h)
struct cara_g8_product {
  const void* X; const void* Gt;
  float* slabs; float* cs_slabs;
  int ldx, K1, nchunks, nblk;
};
struct cara_g8_riders {
  cara_g8_product a, b;
  int ldg, M, any_cs, nt;
};

// CARA_OK when the product was launched, CARA_E_LAUNCH on a failed launch, -1 when this tile does not take the
// product (the caller then runs the 128 x 128 x 32 kernel).  mt = rows per tile: 160 or 256.  ts: the pair of
// transposed skinny products the launch carries behind its tiles, or NULL.
int cara_gemm8_launch(const cara_gemm_args* a, hipStream_t st, int mt, const cara_g8_riders* ts);
// what cara_gemm8_launch would do with the product: 0 = not taken; 1 = taken, riding products (riders_nt = their column tiles of
// 16, 0 = none) as workgroups behind the tiles: one slab per tskinny BLOCK; 2 = taken with helper waves: the riding products write
// one slab per WAVE (cara_ts_reduce::wave_slabs)
int cara_gemm8_plan(const cara_gemm_args* a, int mt, int riders_nt);

// helper waves on (CARA_GEMM8_HELPERS=1 or the debug setter; off by default): only then does a riding product write one slab per WAVE
bool cara_gemm8_helpers_on();

// the dispatcher's policy (gemm.hip): does a product of this shape go to the tile?  riders: the launch carries transposed skinny
// products.  (Callers that choose activation layouts ask.)
bool cara_gemm8_policy(int M, int N, int K, int riders);

// skinny.hip: cara_tskinny_partial2_r at rank <= 16, Rp = 32 with a two-stage ring (40 KiB per block)
int cara_tskinny_partial2_small(const void* Xa, int ldxa, const void* Gta, void* slabs_a, int K1a, const void* Xb, int ldxb, const void* Gtb,
                                void* slabs_b, int K1b, int want_colsum_b, int ldg, int M, int Rp, int rank, void* stream);
