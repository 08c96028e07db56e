// The MT x 256 x 64 one-workgroup-per-CU GEMM tile (gemm8.hip) as seen by the dispatcher in gemm.hip.
#pragma once
#include "common.h"

// CARA_OK when the product was launched, CARA_E_LAUNCH on a failed launch, -1 when this tile does not take the
// product (the caller then runs the 128 x 128 x 32 kernel).  mt = rows per tile: 160 or 256.
int cara_gemm8_launch(const cara_gemm_args* a, hipStream_t st, int mt);
