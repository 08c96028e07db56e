#!/usr/bin/env python3
"""Probe: would two half-batches on two HIP streams (kernels of one half filling the tails / the HBM-bound phases
of the other) beat one full batch on one stream?  Two independent engines of batch B/2 enqueued alternately on
two streams, against one engine of batch B.  Measurement only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


KEEP = []


def make(batch, seed):
    model, trainable = bench.build_model(16, 0.1, 100, dev, seed=14, name="vit_base_patch16_224_in21k")
    opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 100, (batch,), generator=g).to(dev)
    KEEP.append(model)   # the engine holds its model weakly
    return model._cara_engine, opt, x, y


def timed(fn, n=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


e1, o1, x1, y1 = make(B, 1)
ms_full = timed(lambda: e1.train_step(x1, y1, o1))
print(f"one stream, batch {B}: {ms_full:.3f} ms/step = {B / ms_full * 1e3:.0f} img/s")
ea, oa, xa, ya = make(B // 2, 2)
eb, ob, xb, yb = make(B // 2, 3)
ms_half = timed(lambda: ea.train_step(xa, ya, oa))
print(f"one stream, batch {B // 2}: {ms_half:.3f} ms/step = {B // 2 / ms_half * 1e3:.0f} img/s")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa):
        ea.train_step(xa, ya, oa)
    with torch.cuda.stream(sb):
        eb.train_step(xb, yb, ob)


ms_two = timed(both)
print(f"two streams, 2 x batch {B // 2}: {ms_two:.3f} ms per pair = {B / ms_two * 1e3:.0f} img/s")
