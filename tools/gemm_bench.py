#!/usr/bin/env python3
"""Micro-benchmark of cara_gemm_bf16 on the GEMM shapes of one adapted ViT-B block (bs 64):
random bf16 operands (never zeros: MI355X clocks higher on zero data), HIP-event timing of
back-to-back launches in one process.  Usage: python tools/gemm_bench.py [--iters 30] [--only NAME]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402

M0 = 64 * 197
SHAPES = [  # name, N, K, epi
    ("qkv_fwd", 2304, 768, "bf16"), ("proj_fwd", 768, 768, "resid"), ("fc1_fwd", 3072, 768, "gelu"),
    ("fc2_fwd", 768, 3072, "resid"), ("fc2_bwd", 3072, 768, "dgelu"), ("fc1_bwd", 768, 3072, "bf16"),
    ("proj_bwd", 768, 768, "bf16"), ("qkv_bwd", 768, 2304, "bf16"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default=None)
    ap.add_argument("--rp", type=int, default=32)
    ap.add_argument("--shape", action="append", default=[],
                    help="M,N,K[,epi] instead of the block shapes (repeatable), e.g. 4096,4096,768,bf16")
    ap.add_argument("--sk", action="store_true", help="give the GEMM stream-K scratch (persistent kernel)")
    ap.add_argument("--packed", action="store_true", help="give the GEMM the K-panel-major image of B as well (Bp)")
    ap.add_argument("--apanels", action="store_true", help="A in K-panel-major layout as well (a_panels)")
    ap.add_argument("--blas", action="store_true",
                    help="also time torch.matmul (hipBLASLt, plain GEMM, no adapter columns, no epilogue) on the "
                         "same operands: a known-good ceiling for these shapes, measurement only")
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    tot_t = tot_f = 0.0
    scratch = torch.zeros(L.gemm_scratch_bytes(), dtype=torch.uint8, device=dev) if args.sk else None
    shapes = [(n, M0, N, K, e) for n, N, K, e in SHAPES]
    if args.shape:
        shapes = []
        for sp in args.shape:
            f = sp.split(",")
            shapes.append((sp, int(f[0]), int(f[1]), int(f[2]), f[3] if len(f) > 3 else "bf16"))
    for name, M, N, K, epi in shapes:
        if args.only and args.only != name:
            continue
        A = torch.randn(M, K, generator=g).bfloat16().to(dev)
        B = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(dev)
        A2 = torch.randn(M, args.rp, generator=g).bfloat16().to(dev)
        B2 = (torch.randn(N, args.rp, generator=g) * 0.02).bfloat16().to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        kw = dict(A2=A2, B2=B2, bias=bias)
        if args.packed:
            kw["Bp"] = L.pack_b_panels(B)
        Arow = A
        if args.apanels:
            A = A.view(M, K // 32, 32).permute(1, 0, 2).contiguous()
            kw.update(a_panels=M, M=M, K=K, lda=K)
        if args.sk:
            kw["scratch"] = scratch
        if epi == "bf16":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw.update(epi=L.EPI_BF16)
        elif epi == "gelu":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw.update(epi=L.EPI_GELU, C2=torch.empty_like(out))
        elif epi == "dgelu":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw.update(epi=L.EPI_DGELU, aux=torch.randn(M, N, generator=g).bfloat16().to(dev))
            kw["bias"] = None
        else:
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
            kw.update(epi=L.EPI_RESID, aux=torch.randn(M, N, generator=g).to(dev),
                      rowscale=torch.ones((M + 196) // 197, device=dev), rows_per_sample=197)
        for _ in range(3):
            L.gemm(A, B, out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            L.gemm(A, B, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = 2.0 * M * N * (K + 16)
        tot_t += us
        tot_f += fl
        extra = ""
        if args.blas:
            Bt = B.t()
            for _ in range(3):
                torch.matmul(Arow, Bt)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                torch.matmul(Arow, Bt)
            e1.record()
            torch.cuda.synchronize()
            ub = e0.elapsed_time(e1) * 1e3 / args.iters
            extra = f"   | hipBLASLt plain {ub:8.1f} us {2.0 * M * N * K / ub / 1e6:7.1f} TF/s"
        print(f"{name:9s} N={N:5d} K={K:5d} {epi:6s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s{extra}", flush=True)
    if tot_t:
        print(f"block total {tot_t:8.1f} us  {tot_f / tot_t / 1e6:7.1f} TF/s  (x12 layers = {12 * tot_t / 1e3:.2f} ms/step)")


if __name__ == "__main__":
    main()
