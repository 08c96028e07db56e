#!/usr/bin/env python3
"""Yardstick for the GEMM family: TF/s alone, on RANDOM operands, of
  * the product's 128x128x32 / 160x128x32 kernel (gemm.hip, frozen weights from their K-panel-major image),
  * the MT x 256 x 64 one-workgroup-per-CU tile of gemm8.hip at MT = 160 and MT = 256,
  * hipBLASLt's plain product (torch.matmul: no adapter columns, no epilogue) -- a known-good ceiling,
at 4096^3 and on the eight products of an adapted ViT-B block (M = 64 * 197), all variants interleaved in ONE process
(rounds x variants; median and min per variant).  Also checks gemm8 against the product kernel bit for bit.
Usage: python tools/gemm8_bench.py [--iters 20] [--rounds 5] [--only NAME]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402

M0 = 64 * 197
SHAPES = [  # name, M, N, K, epi, Rp
    ("4096^3", 4096, 4096, 4096, "bf16", 0),
    ("qkv_fwd", M0, 2304, 768, "bf16", 32), ("proj_fwd", M0, 768, 768, "resid", 32), ("fc1_fwd", M0, 3072, 768, "gelu", 32),
    ("fc2_fwd", M0, 768, 3072, "resid", 32), ("fc2_bwd", M0, 3072, 768, "dgelu", 32), ("fc1_bwd", M0, 768, 3072, "bf16", 32),
    ("proj_bwd", M0, 768, 768, "bf16", 32), ("qkv_bwd", M0, 768, 2304, "bf16", 32),
]


def set_g8(mt):
    L.lib().cara_debug_set_gemm8(int(mt))


def build(name, M, N, K, epi, Rp, g, dev):
    A = torch.randn(M, K, generator=g).bfloat16().to(dev)
    B = ((torch.rand(N, K, generator=g) * 2 - 1) * (0.05 if M != N else 1.0)).bfloat16().to(dev)
    kw = {}
    if Rp:
        kw.update(A2=torch.randn(M, Rp, generator=g).bfloat16().to(dev), B2=(torch.randn(N, Rp, generator=g) * 0.02).bfloat16().to(dev))
    if epi != "dgelu" and M != N:
        kw["bias"] = torch.randn(N, generator=g).to(dev)
    if epi == "bf16":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw.update(epi=L.EPI_BF16)
    elif epi == "gelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw.update(epi=L.EPI_GELU, C2=torch.empty_like(out))
    elif epi == "dgelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw.update(epi=L.EPI_DGELU, aux=torch.randn(M, N, generator=g).bfloat16().to(dev))
    else:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
        kw.update(epi=L.EPI_RESID, aux=torch.randn(M, N, generator=g).to(dev),
                  rowscale=(torch.rand((M + 196) // 197, generator=g) + 0.5).to(dev), rows_per_sample=197)
    return A, B, out, kw


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':9s} {'N':>5s} {'K':>5s} {'epi':6s} | variant: median us (min) TF/s", flush=True)
    for name, M, N, K, epi, Rp in SHAPES:
        if args.only and args.only != name:
            continue
        A, B, out, kw = build(name, M, N, K, epi, Rp, g, dev)
        Bp = L.pack_b_panels(B)
        variants = {}
        # the product kernel as the model runs it (weights from the K-panel-major image)
        variants["product"] = (0, dict(kw, Bp=Bp), out)
        variants["g8-160"] = (160, kw, torch.empty_like(out))
        if Rp == 0:
            variants["g8-256"] = (256, kw, torch.empty_like(out))
        # plain products (no adapter columns, bf16 epilogue, no bias): what hipBLASLt's number compares with
        plain_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if Rp:
            variants["g8-160 plain"] = (160, dict(epi=L.EPI_BF16), plain_out)
            variants["g8-256 plain"] = (256, dict(epi=L.EPI_BF16), torch.empty_like(plain_out))
            variants["product plain"] = (0, dict(epi=L.EPI_BF16, Bp=Bp), torch.empty_like(plain_out))
        Bt = B.t()
        fns = {}
        for vn, (mt, vkw, o) in variants.items():
            def fn(mt=mt, vkw=vkw, o=o):
                set_g8(mt)
                L.gemm(A, B, o, **vkw)
            fns[vn] = fn
        fns["hipBLASLt plain"] = lambda: torch.matmul(A, Bt)
        # correctness: gemm8 == the product kernel, bit for bit (same order of the 32-deep MFMA steps per accumulator)
        for vn, fn in fns.items():
            fn()
        torch.cuda.synchronize()
        ref = variants["product"][2]
        for vn in ("g8-160", "g8-256"):
            if vn in variants:
                o = variants[vn][2]
                same = torch.equal(o, ref)
                md = (o.float() - ref.float()).abs().max().item()
                print(f"   check {vn} vs product: {'bitwise equal' if same else 'DIFFERENT max|d| = %.3g' % md}", flush=True)
        if "g8-256 plain" in variants:
            o, r2 = variants["g8-256 plain"][2], variants["product plain"][2]
            print(f"   check g8-256 plain vs product plain: {'bitwise equal' if torch.equal(o, r2) else 'DIFFERENT'}", flush=True)
        times = {vn: [] for vn in fns}
        for _ in range(args.rounds):
            for vn, fn in fns.items():
                fn()
                times[vn].append(timed(fn, args.iters))
        set_g8(-1)
        for vn, ts in times.items():
            fl = 2.0 * M * N * (K + (Rp if (vn in ("product", "g8-160", "g8-256") and Rp) else 0))
            med, mn = statistics.median(ts), min(ts)
            print(f"{name:9s} {N:5d} {K:5d} {epi:6s} | {vn:16s} {med:8.1f} us ({mn:8.1f})  {fl / med / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
