#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (no synchronisation), against the GPU's time per step?
If the two are close the step is launch-bound and a hipGraph (or fewer launches) is what helps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
model, trainable = bench.build_model(16, 0.1, 100, dev, seed=14, name="vit_base_patch16_224_in21k")
eng = model._cara_engine
if os.environ.get("CARA_BENCH_TORCH_ADAMW") == "1":
    opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4, fused=True)
else:
    from cara_amd.optim import AdamW
    opt = AdamW(trainable, lr=1e-3, weight_decay=1e-4)
x = torch.randn(64, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (64,), device=dev)
for _ in range(5):
    eng.train_step(x, y, opt)
torch.cuda.synchronize()
host, parts = [], []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter()
    eng.train_step(x, y, opt)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step: mean {1e3*sum(host)/len(host):.3f} ms (min {1e3*min(host):.3f}, max {1e3*max(host):.3f}); "
      f"loop without sync {1e3*(t1-t0)/20:.3f} ms/step; with final sync {1e3*(t2-t0)/20:.3f} ms/step")
# split of the host time: forward / backward / optimizer
import cara_amd.engine as E
orig_f, orig_b = eng._run_forward, eng._run_backward
acc = {"f": 0.0, "b": 0.0, "o": 0.0}
def tf(*a, **k):
    s = time.perf_counter(); r = orig_f(*a, **k); acc["f"] += time.perf_counter() - s; return r
def tb(*a, **k):
    s = time.perf_counter(); r = orig_b(*a, **k); acc["b"] += time.perf_counter() - s; return r
eng._run_forward, eng._run_backward = tf, tb
ostep = opt.step
def to(*a, **k):
    s = time.perf_counter(); r = ostep(*a, **k); acc["o"] += time.perf_counter() - s; return r
opt.step = to
torch.cuda.synchronize()
for _ in range(20):
    eng.train_step(x, y, opt)
torch.cuda.synchronize()
print({k: round(1e3 * v / 20, 3) for k, v in acc.items()}, "ms per step on the host (forward / backward / optimizer)")
