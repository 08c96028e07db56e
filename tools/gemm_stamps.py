#!/usr/bin/env python3
"""Timeline of ONE GEMM launch from in-kernel time stamps (diagnostic build of the library with -DCARA_GEMM_STAMPS,
tools/build_stamps.sh): every workgroup records s_memrealtime (100 MHz) at its start, at the end of its K loop and when
its epilogue's stores have retired, and the CU it ran on.  Prints how long K loops and epilogues take, how many
workgroups are in which phase over time (are the rounds in phase?), and one CU's schedule.

  CARA_LIB_PATH=tools/probe/libcara_stamps.so python tools/gemm_stamps.py --shape fc1_fwd
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cara_amd import _lib as L  # noqa: E402

M0 = 64 * 197
SHAPES = {"qkv_fwd": (2304, 768, "bf16"), "fc1_fwd": (3072, 768, "gelu"), "fc2_bwd": (3072, 768, "dgelu"),
          "fc2_fwd": (768, 3072, "resid"), "fc1_bwd": (768, 3072, "bf16"), "proj_fwd": (768, 768, "resid")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fc1_fwd")
    ap.add_argument("--bin", type=float, default=2.0, help="histogram bin, us")
    ap.add_argument("--ts", action="store_true", help="the launch also carries the linear's two transposed skinny products (cara_gemm_with_tskinny)")
    ap.add_argument("--cold", action="store_true", help="write 512 MB to another buffer before the measured launch")
    a = ap.parse_args()
    N, K, epi = SHAPES[a.shape]
    M = M0
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K, generator=g).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(dev)
    kw = dict(A2=torch.randn(M, 32, generator=g).bfloat16().to(dev), B2=(torch.randn(N, 32, generator=g) * 0.02).bfloat16().to(dev),
              bias=torch.randn(N, generator=g).to(dev), Bp=L.pack_b_panels(B))
    if epi == "bf16":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); kw.update(epi=L.EPI_BF16)
    elif epi == "gelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); kw.update(epi=L.EPI_GELU, C2=torch.empty_like(out))
    elif epi == "dgelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw.update(epi=L.EPI_DGELU, aux=torch.randn(M, N, generator=g).bfloat16().to(dev)); kw["bias"] = None
    else:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
        kw.update(epi=L.EPI_RESID, aux=torch.randn(M, N, generator=g).to(dev), rowscale=torch.ones(64, device=dev), rows_per_sample=197)
    lib = L.lib()
    nmax = 8192
    buf = torch.zeros(nmax * 4, dtype=torch.int64, device=dev)
    lib.cara_debug_gemm_stamps.argtypes = [C.c_void_p]
    junk = torch.empty(512 * 1000 * 1000 // 4, device=dev)
    launch = lambda: L.gemm(A, B, out, **kw)
    if a.ts:   # dX = dY Wt^T with dU = X^T G' and dVs = dY^T T riding: A = dY [M, K], X = the linear's input [M, N]
        lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
        X = torch.randn(M, N, generator=g).bfloat16().to(dev)
        ldg = (M + 31) // 32 * 32
        Gt = torch.zeros(32, ldg, dtype=torch.bfloat16, device=dev); Tt = torch.zeros_like(Gt)
        Gt[:, :M] = torch.randn(32, M, generator=g).bfloat16().to(dev); Tt[:, :M] = torch.randn(32, M, generator=g).bfloat16().to(dev)
        sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, N, 32)), dtype=torch.uint8, device=dev)
        sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, K, 32)), dtype=torch.uint8, device=dev)
        ga = L.GemmArgs()
        ga.A, ga.lda, ga.B, ga.ldb, ga.Bp, ga.A2, ga.B2, ga.Rp = L.ptr(A), K, L.ptr(B), K, L.ptr(kw["Bp"]), L.ptr(kw["A2"]), L.ptr(kw["B2"]), 32
        ga.M, ga.N, ga.K, ga.C, ga.ldc, ga.epi = M, N, K, L.ptr(out), N, kw["epi"]
        ga.aux = L.ptr(kw.get("aux"))
        keep = (X, Gt, Tt, sa, sb)
        launch = lambda: L.check(lib.cara_gemm_with_tskinny(C.byref(ga), L.ptr(X), N, L.ptr(Gt), L.ptr(sa), N, L.ptr(A), K, L.ptr(Tt), L.ptr(sb), K,
                                                           1, ldg, M, 32, L.stream()), "cara_gemm_with_tskinny")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    assert lib.cara_debug_gemm_stamps(C.c_void_p(buf.data_ptr())) == 0
    if a.cold:
        junk.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    lib.cara_debug_gemm_stamps(C.c_void_p(0))
    s = buf.cpu().view(nmax, 4)
    used = (s[:, 2] != 0).nonzero().flatten()
    s = s[used]
    t0 = int(s[:, 0].min())
    st, kl, en = [(s[:, i] - t0).double() / 100.0 for i in range(3)]   # us
    hw = s[:, 3]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    print(f"{a.shape}: N={N} K={K} {epi}; {len(used)} workgroups on {len(set(cu.tolist()))} CUs; event time {e0.elapsed_time(e1) * 1e3:.1f} us; "
          f"span of stamps {float(en.max()):.1f} us")
    if a.ts:   # the products' blocks sit behind the GEMM's tiles
        th = 160 if N >= 3072 else 128
        ntile = ((M + th - 1) // th) * (N // 128)
        idx = used
        r = idx >= ntile
        print(f"GEMM tiles {int((~r).sum())}: last K-loop end {float(kl[~r].max()):.1f} us, last end {float(en[~r].max()):.1f} us;  "
              f"product blocks {int(r.sum())}: start {float(st[r].min()):.1f} .. {float(st[r].max()):.1f} us, end {float(en[r].min()):.1f} .. {float(en[r].max()):.1f} us, "
              f"duration median {float((en[r] - st[r]).median()):.1f} us")
    kd, ed = kl - st, en - kl
    q = lambda x: [round(float(v), 1) for v in torch.quantile(x, torch.tensor([0.05, 0.5, 0.95], dtype=torch.double))]
    print(f"K loop us (5/50/95 %): {q(kd)}   epilogue us: {q(ed)}   start us: {q(st)}")
    nb = int(float(en.max()) / a.bin) + 1
    print("  t(us)  in K loop  in epilogue   started  finished")
    for b in range(nb):
        lo, hi = b * a.bin, (b + 1) * a.bin
        mid = (lo + hi) / 2
        ink = int(((st <= mid) & (kl > mid)).sum()); ine = int(((kl <= mid) & (en > mid)).sum())
        print(f"  {lo:5.0f}  {ink:9d}  {ine:11d}  {int(((st >= lo) & (st < hi)).sum()):8d}  {int(((en >= lo) & (en < hi)).sum()):8d}")
    # the busiest CU's schedule
    ids, cnt = torch.unique(cu, return_counts=True)
    c = int(ids[cnt.argmax()])
    rows = sorted((float(st[i]), float(kl[i]), float(en[i])) for i in range(len(cu)) if int(cu[i]) == c)
    print(f"CU {c}: {len(rows)} workgroups (start, K loop end, end):")
    for r in rows:
        print(f"   {r[0]:7.1f} {r[1]:7.1f} {r[2]:7.1f}")


if __name__ == "__main__":
    main()
