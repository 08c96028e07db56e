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

from ._lib import CaraError


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


def checkpoint_name(dataset: str, acc: float, seed: int, directory: str = ".") -> str:
    """File name of the reference's best-accuracy checkpoint (vit_cp.py:65)."""
    import os
    return os.path.join(directory, f"vit_{dataset}_{round(acc, 5)}_seed_{seed}.pt")


def save_checkpoint(model, path: str) -> None:
    """vit_cp.py:66: ``th.save(vit.state_dict(), name)`` -- the WHOLE state dict (frozen backbone, 12 CP_* tensors,
    head), keys of timm 0.4.12, tensors on the CPU so that the file loads anywhere."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_checkpoint(model, path: str) -> None:
    """vit_cp.py:170 (``--evaluate``): ``vit.load_state_dict(th.load(path))``, strict, after ``cara()`` and
    ``reset_classifier()``.  The engine notices the in-place change of the backbone parameters and re-ingests them."""
    sd = torch.load(path, map_location="cpu")
    model.load_state_dict(sd)


def evaluate_only(model, path: str, test_batches: Iterable) -> float:
    """The ``--evaluate`` flow of vit_cp.py:168-173: load, test, return the accuracy."""
    load_checkpoint(model, path)
    return evaluate(model, test_batches)


class GraphedTrainStep:
    """``CaraEngine.train_step`` captured into a hipGraph and replayed (``cara_vit_forward`` / ``cara_vit_backward`` only enqueue
    work on the caller's stream, include/cara_hip.h; the optimiser must be ``cara_amd.optim.AdamW(capturable=True)``: its step count
    and learning rates live in device memory).  One graph per (batch shape, train / eval mode): the first call with a new key runs
    eagerly (weights ingested, workspaces sized), the second captures, later ones copy the batch into the graph's input buffers,
    upload { step, lr } and replay.  What a capture cannot hold runs eagerly, every time: more than one rank (the all-reduce) and the
    exact weight-dropout mode (a fresh host-side seed per step).  Same switch as ``bench.py --graph``."""

    def __init__(self, engine, optimizer):
        if not getattr(optimizer, "capturable", False):
            raise CaraError("GraphedTrainStep needs cara_amd.optim.AdamW(capturable=True)")
        self.eng, self.opt = engine, optimizer
        self._graphs = {}

    def __call__(self, x, y, group=None):
        from . import dist as cdist
        eng, model = self.eng, self.eng._model()
        self.opt.advance()
        if cdist.world_size(group) > 1 or (eng.weight_dropout == "exact" and model.training):
            return eng.train_step(x, y, self.opt, group=group)
        key = (tuple(x.shape), bool(model.training), eng.precision)
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs[key] = "warm"
            return eng.train_step(x, y, self.opt, group=group)
        if ent == "warm":
            xs, ys = x.clone(), y.clone()
            torch.cuda.synchronize(x.device)
            gr = torch.cuda.CUDAGraph()
            gen = eng._device_generator(x.device)
            if gen is not None:
                gr.register_generator_state(gen)
            side = torch.cuda.Stream(x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    loss = eng.train_step(xs, ys, self.opt, group=group)
            torch.cuda.current_stream(x.device).wait_stream(side)
            ent = self._graphs[key] = (gr, xs, ys, loss)
            # (the capture itself ran nothing: fall through to the replay, which is this call's step)
        gr, xs, ys, loss = ent
        xs.copy_(x)
        ys.copy_(y)
        gr.replay()
        return loss


def fit(model, train_batches: Callable[[int], Iterable], test_batches: Optional[Callable[[], Iterable]] = None,
        epochs: int = 100, lr: float = 1e-3, weight_decay: float = 1e-4, group=None, reference_eval_quirk: bool = True,
        on_eval: Optional[Callable[[int, float], None]] = None, save_best: Optional[dict] = None, seed: Optional[int] = None,
        graph: bool = False):
    """``train_batches(epoch)`` yields (images, labels) already on the device (per-rank shard under
    data parallelism).  Returns (best accuracy, optimizer).

    ``save_best = {"dataset": name, "seed": s, "dir": path}``: keep the state dict of the best evaluation so far on
    disk as the reference does (vit_cp.py:61-66: save on improvement, delete the previous file); the path of the
    file is left in ``save_best["path"]``.  Rank 0 only under data parallelism.
    Data parallel (``torch.distributed`` initialised, more than one rank in ``group``): the trainable parameters are
    broadcast from rank 0 once, and every rank draws its own DropPath / weight-dropout masks from private streams
    seeded with (``seed``, rank); ``seed = None`` derives it from ``torch.initial_seed()`` (what the reference's
    ``torch.manual_seed(args.seed)`` set, vit_cp.py:139-144), so runs with different global seeds stay uncorrelated.
    Single process: the masks come from torch's global generators, exactly as in the reference, unless a ``seed`` is
    passed explicitly.
    ``graph = True``: the step is replayed from a hipGraph (``GraphedTrainStep``; the host then issues one launch per step
    instead of ~600: for boxes where several ranks share few cores).  Eager is the default: the step is GPU-bound."""
    from . import dist as cdist
    model.train()
    params = trainable_parameters(model)
    if cdist.world_size(group) > 1:
        cdist.broadcast_parameters(params, group=group)
    if cdist.world_size(group) > 1 or seed is not None:
        run_seed = int(seed) if seed is not None else int(torch.initial_seed() % (1 << 40))
        model._cara_engine.seed_rank_streams(run_seed, cdist.get_rank(group))
    if seed is None:
        seed = int(torch.initial_seed() % (1 << 31))   # (only names the checkpoint file below, like args.seed in vit_cp.py:65)
    # vit_cp.py:185's torch.optim.AdamW, as one HIP launch over the 14 trainable tensors (cara_amd/optim.py: same arithmetic)
    from .optim import AdamW
    opt = AdamW(params, lr=lr, weight_decay=weight_decay, capturable=graph)
    sched = CosineLRScheduler(opt, t_initial=100, warmup_t=10, lr_min=1e-5, warmup_lr_init=1e-6, decay_rate=0.1)
    eng = model._cara_engine
    gstep = GraphedTrainStep(eng, opt) if graph else None
    best = 0.0
    for epoch in range(epochs):
        for x, y in train_batches(epoch):
            if gstep is not None:
                gstep(x, y, group=group)
            else:
                eng.train_step(x, y, opt, group=group)
            if sched is not None:
                sched.step(epoch)
        if epoch % 10 == 0 and epoch != 0:
            if epoch >= 50:
                sched = None
            if test_batches is not None:
                acc = evaluate(model, test_batches())
                if acc > best and save_best is not None and cdist.get_rank(group) == 0:
                    import os
                    old = save_best.get("path")
                    if old and os.path.exists(old):
                        os.remove(old)
                    save_best["path"] = checkpoint_name(save_best.get("dataset", "task"), acc, save_best.get("seed", seed),
                                                        save_best.get("dir", "."))
                    save_checkpoint(model, save_best["path"])
                best = max(best, acc)
                if on_eval:
                    on_eval(epoch, acc)
                if not reference_eval_quirk:
                    model.train()
    return best, opt
