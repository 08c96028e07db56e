"""The reference's fine-tuning recipe as a host-side helper (``image_classification/vit_cp.py``).

* ``CosineLRScheduler``: timm 0.4.12 semantics for the arguments the reference passes
  (``vit_cp.py:187``: t_initial=100, warmup_t=10, lr_min=1e-5, warmup_lr_init=1e-6, decay_rate=0.1,
  single cycle, no warm-up prefix), stepped as the reference does: ``sched.step(epoch)`` after every
  BATCH (``vit_cp.py:55-56``), so the LR is a function of the epoch index only.
* ``fit``: the loop of ``vit_cp.py:19-70`` -- AdamW over the parameters whose name contains "CP" or
  "head" (``:175-185``), evaluation at epochs 10, 20, ... (``:57``), scheduler dropped from epoch
  50 on (``:58-59``), and the reference's quirk that ``test()`` switches to eval mode and nothing
  switches back (``:75``; DropPath is therefore only active for epochs 0-10).
The arithmetic of every step runs in libcara_hip.so through ``CaraEngine.train_step``.
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Optional

import torch


class CosineLRScheduler:
    def __init__(self, optimizer, t_initial: int, lr_min: float = 0.0, warmup_t: int = 0, warmup_lr_init: float = 0.0,
                 decay_rate: float = 1.0, cycle_limit: int = 1):
        self.opt = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.decay_rate, self.cycle_limit = decay_rate, cycle_limit
        self.base = [g["lr"] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(b - warmup_lr_init) / warmup_t for b in self.base]
            self._set([warmup_lr_init] * len(self.base))   # timm initialises the groups to warmup_lr_init
        else:
            self.warmup_steps = [1.0] * len(self.base)

    def _set(self, lrs):
        for g, lr in zip(self.opt.param_groups, lrs):
            g["lr"] = lr

    def lr_at(self, t: int):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        gamma = self.decay_rate ** i
        if i < self.cycle_limit:
            return [self.lr_min * gamma + 0.5 * (b * gamma - self.lr_min * gamma) * (1 + math.cos(math.pi * t_curr / self.t_initial))
                    for b in self.base]
        return [self.lr_min * (self.decay_rate ** self.cycle_limit)] * len(self.base)

    def step(self, epoch: int):
        self._set(self.lr_at(epoch))


def trainable_parameters(model):
    """vit_cp.py:175-183: train names containing "CP" or "head", freeze the rest."""
    out = []
    for n, p in model.named_parameters():
        if "CP" in n or "head" in n:
            out.append(p)
        else:
            p.requires_grad = False
    return out


@torch.no_grad()
def evaluate(model, batches: Iterable) -> float:
    """vit_cp.py:73-82: eval mode (and it stays on), top-1 accuracy."""
    model.eval()
    hit = tot = 0
    for x, y in batches:
        out = model(x)
        hit += (out.argmax(dim=1).view(-1) == y).sum().item()
        tot += y.numel()
    return hit / max(tot, 1)


def fit(model, train_batches: Callable[[int], Iterable], test_batches: Optional[Callable[[], Iterable]] = None,
        epochs: int = 100, lr: float = 1e-3, weight_decay: float = 1e-4, group=None, reference_eval_quirk: bool = True,
        on_eval: Optional[Callable[[int, float], None]] = None):
    """``train_batches(epoch)`` yields (images, labels) already on the device (per-rank shard under
    data parallelism).  Returns (best accuracy, optimizer)."""
    model.train()
    params = trainable_parameters(model)
    try:
        opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, fused=True)
    except Exception:
        opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    sched = CosineLRScheduler(opt, t_initial=100, warmup_t=10, lr_min=1e-5, warmup_lr_init=1e-6, decay_rate=0.1)
    eng = model._cara_engine
    best = 0.0
    for epoch in range(epochs):
        for x, y in train_batches(epoch):
            eng.train_step(x, y, opt, group=group)
            if sched is not None:
                sched.step(epoch)
        if epoch % 10 == 0 and epoch != 0:
            if epoch >= 50:
                sched = None
            if test_batches is not None:
                acc = evaluate(model, test_batches())
                best = max(best, acc)
                if on_eval:
                    on_eval(epoch, acc)
                if not reference_eval_quirk:
                    model.train()
    return best, opt
