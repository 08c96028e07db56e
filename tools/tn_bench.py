#!/usr/bin/env python3
"""dW = dY^T X two ways, timed alone (HIP events, 50 launches each): cara_gemm_tn_f32 on the row-major operands vs the batched
cara_gemm_bf16 on transposed copies (the transposes themselves timed separately).  python tools/tn_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = "cuda"
M = 64 * 197
g = torch.Generator().manual_seed(0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for out, inn in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    dY = torch.randn(M, out, generator=g).bfloat16().to(dev)
    X = torch.randn(M, inn, generator=g).bfloat16().to(dev)
    nslab = 4
    dW = torch.empty(nslab, out, inn, device=dev)
    st = L.stream
    t_tn = timeit(lambda: L.check(lib.cara_gemm_tn_f32(L.ptr(dY), out, L.ptr(X), inn, L.ptr(dW), inn, out, inn, M, nslab, C.c_size_t(out * inn), st()), "tn"))
    ldk = (M + 64 * nslab - 1) // (64 * nslab) * (64 * nslab)
    dYt = torch.zeros(out, ldk, dtype=torch.bfloat16, device=dev)
    Xt = torch.zeros(inn, ldk, dtype=torch.bfloat16, device=dev)
    t_tr = timeit(lambda: (L.check(lib.cara_transpose_bf16_ld(L.ptr(dY), C.c_long(out), L.ptr(dYt), C.c_long(ldk), M, out, st()), "t"),
                           L.check(lib.cara_transpose_bf16_ld(L.ptr(X), C.c_long(inn), L.ptr(Xt), C.c_long(ldk), M, inn, st()), "t")))
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.M, a.N, a.K = L.ptr(dYt), ldk, L.ptr(Xt), ldk, out, inn, ldk // nslab
    a.epi, a.C, a.ldc, a.batch = L.EPI_F32, L.ptr(dW), inn, nslab
    a.strideA, a.strideB, a.strideC = ldk // nslab, ldk // nslab, out * inn
    t_nt = timeit(lambda: L.check(lib.cara_gemm_bf16(C.byref(a), st()), "nt"))
    gf = 2.0 * M * out * inn / 1e9
    print(f"out {out:5d} in {inn:5d}: TN {t_tn:7.1f} us ({gf / t_tn / 1e3:6.1f} TF/s)   NT {t_nt:7.1f} us ({gf / t_nt / 1e3:6.1f} TF/s) + transposes {t_tr:6.1f} us")
