#!/usr/bin/env python3
"""Can a loop-bound GEMM (N = 768, K = 3072: small output) and an epilogue-bound GEMM (N = 3072, K = 64, GELU: 155 MB of
output) overlap on the chip?  Times each alone and both together on two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cara_amd import _lib as L
dev = "cuda"
g = torch.Generator().manual_seed(0)
M = 12608
def mk(N, K, epi):
    A = torch.randn(M, K, generator=g).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(epi=epi, Bp=L.pack_b_panels(B))
    if epi == L.EPI_GELU:
        kw["C2"] = torch.empty_like(out)
    return A, B, out, kw
loop = mk(768, 3072, L.EPI_BF16)
epi = mk(3072, 64, L.EPI_GELU)
both_k = mk(3072, 768, L.EPI_GELU)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
def one(t):
    A, B, out, kw = t
    L.gemm(A, B, out, **kw)
def together():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); one(loop); e1 = torch.cuda.Event(); e1.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev); one(epi); e2 = torch.cuda.Event(); e2.record()
    torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)
def serial():
    one(loop); one(epi)
print(f"loop-bound alone  {run(lambda: one(loop)):7.1f} us")
print(f"epilogue-bound alone {run(lambda: one(epi)):7.1f} us")
print(f"serial (one stream)  {run(serial):7.1f} us")
print(f"two streams          {run(together):7.1f} us   (includes ~15 us of event plumbing)")
print(f"fc1 fwd K=768 gelu   {run(lambda: one(both_k)):7.1f} us")
