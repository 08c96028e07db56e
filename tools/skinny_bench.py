#!/usr/bin/env python3
"""cara_skinny_xu alone (T = X U, the HBM-bound adapter contraction): microseconds and TB/s for the shapes of the step.
   python tools/skinny_bench.py
Round-3 finding (profiles/r03_f_skinny_bench.txt): 128, 197 and 256 workgroups of 64 rows take the same 22-24 us, 243 workgroups
of 52 rows too -- a launch is dominated by what every workgroup pays once (its 196 KiB of Ut fragments + the first row group, at
the ~35 GB/s one 16-wave workgroup draws), not by rows per workgroup or by how many CUs take part."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402


def main():
    dev = "cuda"
    lib = L.lib()
    p, st = L.ptr, L.stream
    g = torch.Generator().manual_seed(0)
    for M, K in ((12608, 3072), (12608, 2304), (12608, 768), (16384, 3072), (8192, 3072)):
        X = torch.randn(M, K, generator=g).bfloat16().to(dev)
        Ut = torch.randn(32, K, generator=g).bfloat16().to(dev)
        ldt = (M + 63) // 64 * 64
        T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        Tt = torch.empty(32, ldt, dtype=torch.bfloat16, device=dev)
        X2 = X.clone()
        times = []
        for it in range(12):
            X.copy_(X2)                        # as in the step: the operand has just been written by the kernel in front
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            L.check(lib.cara_skinny_xu(p(X), K, p(Ut), p(T), p(Tt), ldt, M, K, 32, st()), "skinny")
            b.record()
            torch.cuda.synchronize()
            if it >= 2:
                times.append(a.elapsed_time(b) * 1e3)
        times.sort()
        us = times[len(times) // 2]
        print(f"M {M:6d} K {K:5d}: {us:7.1f} us  {M * K * 2 / us / 1e6:5.2f} TB/s   (blocks of 64 rows: {(M + 63) // 64})")


if __name__ == "__main__":
    main()
