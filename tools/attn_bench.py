#!/usr/bin/env python3
"""Micro-benchmark of cara_attention_fwd / _bwd at the headline shape (B 64, N 197, H 12, head dim 64) or
--shape B,N,H: random bf16 qkv, HIP-event timing of back-to-back launches."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,197,12")
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    B, N, H = (int(v) for v in a.shape.split(","))
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).bfloat16().to(dev)
    dout = torch.randn(B * N, H * 64, generator=g).bfloat16().to(dev)
    out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=dev)
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B, H, N, device=dev)
    lib, p, st = L.lib(), L.ptr, L.stream
    sc = C.c_float(64 ** -0.5)

    def fwd():
        L.check(lib.cara_attention_fwd(p(qkv), p(out), p(lse), B, N, H, sc, st()), "fwd")

    def bwd():
        L.check(lib.cara_attention_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), B, N, H, sc, st()), "bwd")

    fl = 4.0 * B * H * N * N * 64
    for name, fn, mult in (("fwd", fwd, 1.0), ("bwd", bwd, 2.5)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"attention {name} B={B} N={N} H={H}: {us:8.1f} us  {mult * fl / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
