#!/usr/bin/env python3
"""Where a head's time goes in the fused attention backward (diagnostic build with -DCARA_ATTN_STAMPS:
tools/build_variant.sh attnstamps -DCARA_ATTN_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at
  0 loop top | 1 after the delta step (T1) | 2 end of the dK/dV sweep | 3 after T2 (K, V images landed, all waves)
  4 dK/dV stores issued, Q/dO rows read | 5 after T3 | 6 end of the dQ sweep | 7 after the wait for the next head's images
for each of the heads it walks.

  CARA_LIB_PATH=tools/probe/libcara_attnstamps.so python tools/attn_stamps.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402


def fwd_stamps(lib, p, st, qkv, out, lse, B, N, H, scale, dev):
    """the persistent forward: 0 loop top | 1 after the two barriers (this head's K, V, Q landed) | 2 S^T done | 3 row max done
    | 4 exponentials + P V done | 5 stores issued"""
    buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    assert lib.cara_debug_attn_stamps(p(buf)) == 0
    for _ in range(3):
        L.check(lib.cara_attention_fwd(p(qkv), p(out), p(lse), B, N, H, C.c_float(scale), st()), "fwd")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    L.check(lib.cara_attention_fwd(p(qkv), p(out), p(lse), B, N, H, C.c_float(scale), st()), "fwd")
    b.record()
    torch.cuda.synchronize()
    print(f"\nforward launch {a.elapsed_time(b) * 1e3:.1f} us")
    t = buf.cpu().reshape(256, 4, 8).double() / 100.0
    t0 = t[:, 0, 0].min()
    names = ["barriers + wait for K, V, Q", "S^T = K Q^T (28 MFMAs)", "mask + row max", "exp + P V (28 MFMAs)", "normalise + stores"]
    for slot in range(3):
        d = t[:, slot, 1:6] - t[:, slot, 0:5]
        print(f"head slot {slot}: starts {float((t[:, slot, 0] - t0).mean()):6.2f} us")
        for i, n in enumerate(names):
            print(f"    {n:28s} mean {float(d[:, i].mean()):6.2f}  min {float(d[:, i].min()):6.2f}  max {float(d[:, i].max()):6.2f} us")
    print(f"last stamp at {float((t[:, 2, 5] - t0).max()):.2f} us after the first")


def main():
    B, H, N = 64, 12, 197
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).bfloat16().to(dev)
    out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H, N, dtype=torch.float32, device=dev)
    dout = torch.randn(B * N, H * 64, generator=g).bfloat16().to(dev)
    dqkv = torch.empty_like(qkv)
    lib = L.lib()
    p, st = L.ptr, L.stream
    scale = 0.125
    L.check(lib.cara_attention_fwd(p(qkv), p(out), p(lse), B, N, H, C.c_float(scale), st()), "fwd")
    buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    assert lib.cara_debug_attn_stamps(p(buf)) == 0
    for _ in range(3):
        L.check(lib.cara_attention_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), B, N, H, C.c_float(scale), st()), "bwd")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    L.check(lib.cara_attention_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), B, N, H, C.c_float(scale), st()), "bwd")
    b.record()
    torch.cuda.synchronize()
    print(f"launch {a.elapsed_time(b) * 1e3:.1f} us")
    t = buf.cpu().reshape(256, 4, 8).double() / 100.0    # us
    t0 = t[:, 0, 0].min()
    names = ["delta+T0/T1", "dK/dV sweep", "T2 wait+barrier", "dkdv stores + row reads", "T3 barrier", "dQ sweep", "wait next images"]
    for slot in range(3):
        d = t[:, slot, 1:] - t[:, slot, :-1]
        print(f"head slot {slot}: starts {float((t[:, slot, 0] - t0).mean()):6.2f} us (min {float((t[:, slot, 0] - t0).min()):.2f} max {float((t[:, slot, 0] - t0).max()):.2f})")
        for i, n in enumerate(names):
            print(f"    {n:26s} mean {float(d[:, i].mean()):6.2f}  min {float(d[:, i].min()):6.2f}  max {float(d[:, i].max()):6.2f} us")
        if slot < 2:
            gap = t[:, slot + 1, 0] - t[:, slot, 7]
            print(f"    dQ stores -> next loop top  mean {float(gap.mean()):6.2f} us")
    print(f"last stamp at {float((t[:, 2, 7] - t0).max()):.2f} us after the first")
    fwd_stamps(lib, p, st, qkv, out, lse, B, N, H, scale, dev)
    raw = buf.cpu().reshape(256, 4, 8)[:, 3].double()
    clk = (raw[:, 2] - raw[:, 0]) / ((raw[:, 3] - raw[:, 1]) / 100.0)      # shader-clock ticks per us = MHz
    print(f"shader clock over the head loop: mean {float(clk.mean()):.0f} MHz (min {float(clk.min()):.0f}, max {float(clk.max()):.0f})")


if __name__ == "__main__":
    main()
