#!/usr/bin/env python3
"""the train step as two half-batch pipelines on two HIP streams (CaraEngine.train_step_two_streams) against
the one-stream step, same box, interleaved rounds; eager and replayed from a hipGraph (no host in the loop); with the second
pipeline started late by a spin kernel (--lag clocks).  Also checks that the two forms give the same gradients."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--lags", default="0,300000,600000")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, trainable = bench.build_model(16, 0.1, 100, dev, seed=14)
eng = model._cara_engine
eng.seed_rank_streams(7, 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(a.batch, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, 100, (a.batch,), generator=g).to(dev)
model.eval()   # (no DropPath draw: the two forms see the same arithmetic; the masks do not change the kernels' cost)
model.train()
for b in model.blocks:   # DropPath off for the equality check
    for n in ("drop_path", "drop_path1", "drop_path2"):
        if hasattr(b, n) and hasattr(getattr(b, n), "drop_prob"):
            getattr(b, n).drop_prob = 0.0


def one():
    return eng.train_step(x, y, None)


def two(lag=0):
    return eng.train_step_two_streams(x, y, None, lag_cycles=lag)


l1 = one().item()
g1 = eng._flat_grad.clone()
l2 = two().item()
g2 = eng._flat_grad.clone()
torch.cuda.synchronize()
rel = ((g1 - g2).norm() / g1.norm()).item()
print(f"loss one stream {l1:.6f}, two streams {l2:.6f}; flat gradient rel-L2 difference {rel:.2e} (bf16 operands: the halves' row tiles "
      f"and split-K slab boundaries differ, so the sums round differently; the arithmetic is the same)", flush=True)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    return gr.replay


lags = [int(v) for v in a.lags.split(",")]
variants = {"one stream, eager": one}
for lg in lags:
    variants[f"two streams, eager, lag {lg}"] = (lambda lg=lg: two(lg))
try:
    variants["one stream, graph"] = graphed(one)
    for lg in lags:
        variants[f"two streams, graph, lag {lg}"] = graphed(lambda lg=lg: two(lg))
except Exception as exc:   # noqa: BLE001
    print("graph capture failed:", type(exc).__name__, str(exc)[:200], flush=True)
for fn in variants.values():
    for _ in range(3):
        fn()
res = {k: [] for k in variants}
for r in range(a.rounds):
    for k, fn in variants.items():
        res[k].append(timed(fn, a.steps))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:36s} median {v[len(v) // 2]:7.3f} ms  min {v[0]:7.3f}  max {v[-1]:7.3f}   ({a.batch / v[len(v) // 2] * 1e3:7.0f} img/s)", flush=True)
