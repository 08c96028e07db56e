#!/usr/bin/env python3
"""the train step as two half-batch pipelines on two HIP streams (train_step_two_streams below) against
the one-stream step, same box, interleaved rounds; eager and replayed from a hipGraph (no host in the loop); with the second
pipeline started late by a spin kernel (--lag clocks).  Also checks that the two forms give the same gradients."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402
import bench  # noqa: E402
from cara_amd import _lib as L  # noqa: E402
from cara_amd._lib import CaraError, check, ptr, stream  # noqa: E402


def train_step_two_streams(self, images, labels, optimizer=None, group=None, lag_cycles: int = 0):
    """EXPERIMENT (result: profiles/r05_d_two_streams.txt -- 1.7-2.3 % under graph replay, a tie eager: below the 4 % bar, not in the product): the step as two half-batch
    pipelines on two HIP streams, so that one half's LayerNorm / attention / epilogue tails can run under the other half's
    K loops.  cara_vit_forward / cara_vit_backward are stateless and take a stream: each half has its own workspace and
    its own flat gradient buffer; the mean over the batch is the average of the halves' means (dlogits scaled by 1/2 at the
    cross-entropy), the second buffer is added to the first, then the usual all-reduce + optimiser.  bf16, factored adapters."""
    model = self._model()
    dev = images.device
    B = images.shape[0]
    if B % 2 or self.precision != "bf16" or self.weight_dropout != "off":
        raise CaraError("train_step_two_streams: even batch, precision 'bf16', weight_dropout 'off'")
    cp = [getattr(model, "CP_" + n) for n in self.cp_fields]
    hw, hb = model.head.weight, model.head.bias
    from .dist import flat_views, world_size
    with torch.no_grad(), torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        if self.__dict__.get("_two") is None:
            names = [(n, getattr(model, "CP_" + n).shape) for n in self.cp_fields] + [("head_w", hw.shape), ("head_b", hb.shape), ("_found_inf", (1,))]
            self._two = {"streams": (torch.cuda.Stream(dev), torch.cuda.Stream(dev)), "flat2": flat_views(names, dev),
                         "loss": [torch.empty(1 + B // 2, device=dev) for _ in range(2)],
                         "dl": [torch.empty(B // 2, hw.shape[0], device=dev) for _ in range(2)]}
        two = self._two
        droppath = self.draw_droppath(model, B, dev)
        gv0 = self._grad_buffers(model, dev)
        flat1, gv1 = two["flat2"]
        ev_in = torch.cuda.Event()
        ev_in.record(main)
        done = []
        for h, (s_, gv) in enumerate(zip(two["streams"], (gv0, gv1))):
            sl = slice(h * (B // 2), (h + 1) * (B // 2))
            x_h, y_h = images[sl], labels[sl].contiguous()
            dp_h = droppath[:, :, sl].contiguous() if droppath is not None else None
            s_.wait_event(ev_in)
            with torch.cuda.stream(s_):
                self._slot = h
                try:
                    if h == 1 and lag_cycles > 0:   # phase shift of the second pipeline (a spin kernel of that many clocks)
                        torch.cuda._sleep(int(lag_cycles))
                    logits = self._run_forward(x_h, dp_h, hw, hb, cp)
                    check(self._lib().cara_cross_entropy_ex(ptr(logits), ptr(y_h), ptr(two["loss"][h]), ptr(two["dl"][h]), B // 2, logits.shape[1],
                                                            C.c_float(0.5 / world_size(group)), None, None, stream(dev)), "cara_cross_entropy_ex")
                    st = self._ws[self._last_key]
                    gps = L.cp_ptrs(self.cp_fields, [gv[n] for n in self.cp_fields])
                    cps = self._cp_ptrs([t.detach().contiguous() for t in cp])
                    st["shape"].loss_scale, st["shape"].found_inf = None, None
                    check(self._lib().cara_vit_backward(C.byref(st["geom"]), C.byref(st["shape"]), C.byref(self._ingested[1]), C.byref(cps),
                                                        ptr(hw.detach()), ptr(two["dl"][h]), ptr(dp_h), ptr(st["ws"]), C.byref(gps),
                                                        ptr(gv["head_w"]), ptr(gv["head_b"]), stream(dev)), "cara_vit_backward")
                finally:
                    self._slot = 0
                e = torch.cuda.Event()
                e.record(s_)
                done.append(e)
        for e in done:
            main.wait_event(e)
        self._flat_grad.add_(flat1)
        self._bwd_ready = -1
        self._apply_gradients(optimizer, group, prescaled=True)
        return two["loss"][0][0] + two["loss"][1][0]



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
eng._keep_ws = True   # every workspace stays allocated: the variants below alternate, and a workspace freed while a side stream still
                      # runs kernels on it would be handed out again by the caching allocator
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
    return train_step_two_streams(eng, x, y, None, lag_cycles=lag)


l1 = one().item()
g1 = eng._flat_grad.clone()
l2 = two().item() / 2
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
