"""``torch.optim.AdamW`` for the trainable tensors of a fine-tuning step as ONE HIP launch.

The reference builds ``torch.optim.AdamW(trainable, lr=..., weight_decay=...)`` over the parameters whose name contains "CP" or
"head" (``/root/reference/image_classification/vit_cp.py:175-185``) and steps it once per batch (``:50``).  Those are 14 small
tensors (1.2e5 elements at rank 16); torch's fused path takes 42 us per step for them on an MI355X (20 workgroups, launch and
latency bound), ``cara_adamw_step`` 3-4 us.  Same arithmetic as torch (``amsgrad=False``, ``maximize=False``; checked
against ``torch.optim.AdamW`` on the CPU in ``tests/test_kernels_gpu.py::test_adamw_step_against_torch``), same constructor
arguments, same ``param_groups`` (a scheduler that writes ``group["lr"]`` works unchanged), ``state_dict()`` in torch's layout
(``step``, ``exp_avg``, ``exp_avg_sq`` per parameter).  Device tensors only: there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L
from ._lib import CaraError


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 capturable: bool = False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("AdamW: lr, eps, weight_decay >= 0 and betas in [0, 1)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > L.ADAMW_MAX_GROUPS:
            raise CaraError(f"cara_amd.optim.AdamW takes at most {L.ADAMW_MAX_GROUPS} parameter groups")
        b = self.param_groups[0]
        if any(g["betas"] != b["betas"] or g["eps"] != b["eps"] for g in self.param_groups):
            raise CaraError("cara_amd.optim.AdamW: betas and eps are shared by all parameter groups")
        # optional device float: when it is non-zero at the time the launch runs, the step changes nothing (CaraEngine sets it to
        # the step's found-inf word under precision = "fp16").  The host-side step counts still advance: a skipped step moves the
        # bias corrections on by one, which is what torch's fused AdamW does under GradScaler too.
        self.skip_flag = None
        # capturable = True: the step count and the learning rates the kernel uses live in device memory (cara_adamw_args::dyn),
        # so that a captured launch stays correct under hipGraph replay.  step() then only enqueues the launch; advance() -- call it
        # once before every step / graph replay, outside the captured region -- moves the count on and uploads { t, lr per group }
        # (one fill launch for the count, one per learning rate that changed).  All parameters share one step count in this mode.
        self.capturable = bool(capturable)
        self._dyn = None
        self._dyn_step = 0

    def advance(self):
        """capturable mode: next step -- count + 1, current learning rates -> device (call before step() / graph.replay())"""
        if not self.capturable:
            raise CaraError("advance() belongs to AdamW(capturable=True)")
        dev = next(p for g in self.param_groups for p in g["params"]).device
        if self._dyn is None or self._dyn.device != dev:
            self._dyn = torch.zeros(1 + L.ADAMW_MAX_GROUPS, device=dev)
            self._dyn_lr = [None] * L.ADAMW_MAX_GROUPS
        self._dyn_step += 1
        # (fill_ launches carry the value in their arguments: stream-ordered and safe however far the host runs ahead of the device.
        # An asynchronous copy from a temporary host tensor is neither -- the first build of this mode read freed host memory.)
        self._dyn[0].fill_(float(self._dyn_step))
        for gi, g in enumerate(self.param_groups):
            if self._dyn_lr[gi] != float(g["lr"]):
                self._dyn_lr[gi] = float(g["lr"])
                self._dyn[1 + gi].fill_(self._dyn_lr[gi])

    def _init_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0)                      # (torch's layout; a host scalar -- nothing reads it on the device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # (re-checked at every step: param groups may have been added since the constructor ran)
        if len(self.param_groups) > L.ADAMW_MAX_GROUPS:
            raise CaraError(f"cara_amd.optim.AdamW takes at most {L.ADAMW_MAX_GROUPS} parameter groups")
        b = self.param_groups[0]
        if any(g["betas"] != b["betas"] or g["eps"] != b["eps"] for g in self.param_groups):
            raise CaraError("cara_amd.optim.AdamW: betas and eps are shared by all parameter groups")
        # parameters bucketed by their step count (torch keeps one per parameter: a parameter whose grad was None for a while, one
        # added later or a loaded state with mixed steps simply has another count): the bias corrections are per launch
        buckets = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise CaraError("cara_amd.optim.AdamW steps dense fp32 parameters on the GPU (no CPU path)")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise CaraError("cara_amd.optim.AdamW needs contiguous parameters and gradients")
                st = self._init_state(p)
                if self.capturable:
                    if self._dyn is None:
                        raise CaraError("AdamW(capturable=True): call advance() before step()")
                    st["step"] = torch.tensor(float(self._dyn_step))
                    buckets.setdefault(max(self._dyn_step, 1), []).append((p, st, gi))
                    continue
                st["step"] += 1
                buckets.setdefault(int(st["step"].item()), []).append((p, st, gi))
        if not buckets:
            return loss
        dev = next(iter(buckets.values()))[0][0].device
        if any(e[0].device != dev for es in buckets.values() for e in es):
            raise CaraError("cara_amd.optim.AdamW: all parameters on one device")
        b1, b2 = self.param_groups[0]["betas"]
        with torch.cuda.device(dev):
            for step, entries in sorted(buckets.items()):
                for i in range(0, len(entries), L.ADAMW_MAX_TENSORS):
                    part = entries[i:i + L.ADAMW_MAX_TENSORS]
                    a = L.AdamWArgs()
                    for j, (p, st, gi) in enumerate(part):
                        a.t[j] = L.AdamWTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                               p.numel(), gi)
                    a.ntensors, a.step = len(part), step
                    for gi, group in enumerate(self.param_groups):
                        a.lr[gi], a.weight_decay[gi] = group["lr"], group["weight_decay"]
                    a.one_minus_beta1, a.beta2, a.one_minus_beta2, a.eps = 1.0 - b1, b2, 1.0 - b2, self.param_groups[0]["eps"]
                    a.bias_correction1 = 1.0 - b1 ** step
                    a.bias_correction2_sqrt = math.sqrt(1.0 - b2 ** step)
                    a.skip_flag = self.skip_flag.data_ptr() if self.skip_flag is not None else None
                    a.dyn = self._dyn.data_ptr() if self.capturable else None
                    L.check(L.lib().cara_adamw_step(C.byref(a), L.stream(dev)), "cara_adamw_step")
        return loss
