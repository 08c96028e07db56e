#!/usr/bin/env python3
"""Which part of a train step issues hipMemcpy calls (the __amd_rocclr_copyBuffer kernels at the step boundary): run under
rocprofv3 --hip-trace --marker-trace is not needed -- the parts run in separate loops of 20 and the API trace is ordered by time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cara_amd.optim import AdamW

dev = torch.device("cuda:0")
model, tr = bench.build_model(16, 0.1, 100, dev, seed=11, name="vit_base_patch16_224_in21k")
eng = model._cara_engine
opt = AdamW(tr, lr=1e-3, weight_decay=1e-4)
x = torch.randn(64, 3, 224, 224).to(dev)
y = torch.randint(0, 100, (64,)).to(dev)
for _ in range(3):
    eng.train_step(x, y, opt)
torch.cuda.synchronize()
def phase(name, fn, n=20):
    torch.cuda.synchronize()
    time.sleep(0.2)
    print("PHASE", name, time.time_ns(), flush=True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
phase("droppath", lambda: eng.draw_droppath(model, 64, dev))
phase("adamw", lambda: opt.step())
phase("train_step", lambda: eng.train_step(x, y, opt))
print("PHASE end", time.time_ns(), flush=True)
