"""Engine: binds an adapted VisionTransformer to the whole-model entry points of libcara_hip.so.

torch owns device memory, the stream and autograd bookkeeping; every FLOP of the forward and
backward is issued by ``cara_vit_forward`` / ``cara_vit_backward`` (include/cara_hip.h).
Reference call sites this replaces: ``out = model(x)`` and ``loss.backward()`` of
``/root/reference/image_classification/vit_cp.py:46-49`` with the patched forwards of
``/root/reference/src/cara/cara.py:15-95``.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch

from . import _lib as L
from ._lib import CaraError, check, ptr, stream


def _rp(rank: int) -> int:
    return 32 if rank <= 32 else 64


class _VitFn(torch.autograd.Function):
    """logits = ViT+CaRA(images); gradients only into the head and the 12 CP tensors."""

    @staticmethod
    def forward(ctx, eng, images, droppath, head_w, head_b, *cp):
        # (grad mode is always off in here: CaraEngine.forward noted beforehand whether a backward can follow)
        logits = eng._run_forward(images, droppath, head_w, head_b, cp, need_backward=eng._need_backward)
        ctx.eng, ctx.droppath, ctx.shape_key = eng, droppath, eng._last_key
        ctx.save_for_backward(head_w, *cp)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        head_w, *cp = ctx.saved_tensors
        eng = ctx.eng
        if eng._last_key != ctx.shape_key or eng._fwd_serial != eng._bwd_ready:
            raise CaraError("backward must follow its own forward: the activation workspace holds one step")
        g = eng._run_backward(dlogits, ctx.droppath, head_w, cp)
        return (None, None, None, g["head_w"], g["head_b"], *[g[n] for n in eng.cp_fields])


class CaraEngine:
    def __init__(self, model, rank: int, scale: float, cp_length: int = 4):
        self._model = weakref.ref(model)
        self.rank, self.Rp, self.scale = rank, _rp(rank), scale
        # order of the QKV tensorisation (dim_experiment.py:188-207): 4 = src/cara's; 3 and 5 run on the same kernels
        self.cp_length = int(cp_length)
        self.cp_fields = L.cp_fields(self.cp_length)
        # Dropout(0.1) on the materialised dW (cara.py:35,57,81,92).  "off": factored adapters, no weight-space
        # dropout (the fast default; a mask on dW's elements does not factor).  "exact": train-mode forwards run on
        # W_eff = W + keep/(1-p) dW with a fresh mask per step and the adapter gradients come from the dense
        # dW = dY^T X, as in the reference (cara_vit_shape::wd_exact; ~1.4x the step time).  Eval is identical in both.
        self.weight_dropout = "off"
        self.weight_dropout_p = 0.1
        self.weight_dropout_seed = None   # tests pin the mask seed; None = a fresh draw from torch's RNG per forward
        # "bf16" (default): the fast path, bf16 MFMA operands.
        # "fp16": the SAME kernels compiled with IEEE-half MFMA operands (libcara_hip_f16.so: 11 significand bits at the bf16 MFMA
        # rate) for forward AND backward -- logits inside north_star's 1e-3 of the fp32 reference (measured: tests/test_model_gpu.py,
        # DESIGN.md section 2); the backward runs under a static loss scale (FP16_LOSS_SCALE: gradients need half's range, not its
        # precision) that is divided out of the flat fp32 gradient buffer before the all-reduce.  Whole-model calls only.
        self.precision = "bf16"
        self._ingested = None
        self._ingest_sig = None
        self._ws = {}
        self._last_key = None
        self._fwd_serial = 0
        self._bwd_ready = -1
        self._need_backward = True
        self._flat_grad = None
        self._grad_views = None
        self._slot = 0          # workspace slot of the call in flight (0 in the product; tools/two_stream_step.py runs two at once)
        # RNG streams of the stochastic parts (DropPath masks on the device, weight-dropout seeds on the host).
        # None = torch's global generators.  Under data parallelism every rank must draw DIFFERENT masks while the
        # parameters stay identical: seed_rank_streams(seed, rank) after the model is built.
        self._gen_dev = None
        self._gen_cpu = None
        self._gen_seed = None
        self._warned_wd = False
        from .modules import FactorPack
        self._factors = FactorPack(self)

    # Loss scaling of precision = "fp16" (half's range, not its precision, is what gradients lack), entirely on the device:
    # the scale lives in a device float (initially FP16_LOSS_SCALE), cara_cross_entropy_ex multiplies dlogits by it, the kernels
    # that WRITE the final gradients divide it out and raise a found-inf word when a value is not finite; that word sits behind
    # the last gradient in the flat buffer, so the step's one all-reduce carries it to every rank; cara_amp_update then halves the
    # scale (or doubles it after AMP_GROWTH_INTERVAL clean steps: torch.amp.GradScaler's rule) and cara_amd.optim.AdamW skips the
    # update of such a step -- no host synchronisation, no extra pass over the gradients.  A foreign optimiser is stepped behind a
    # host-side check of the word instead (one synchronisation per step, fp16 only).
    FP16_LOSS_SCALE = 1024.0
    AMP_GROWTH, AMP_BACKOFF, AMP_GROWTH_INTERVAL, AMP_MAX_SCALE = 2.0, 0.5, 2000, 65536.0
    DROPPATH_AHEAD = 32   # steps of DropPath masks per draw (draw_droppath)

    def _amp(self, dev):
        """device float[4] = {loss scale, clean steps since the last change, steps skipped so far, unused} (fp16 only)"""
        st = self.__dict__.get("_amp_state")
        if st is None or st.device != dev:
            st = torch.tensor([self.FP16_LOSS_SCALE, 0.0, 0.0, 0.0], device=dev)
            self._amp_state = st
        return st

    @property
    def loss_scale(self) -> float:
        """current loss scale of precision = "fp16" (synchronises; diagnostics)"""
        st = self.__dict__.get("_amp_state")
        return float(st[0].item()) if st is not None else self.FP16_LOSS_SCALE

    @property
    def skipped_steps(self) -> int:
        """steps whose gradients overflowed under the loss scale and were skipped (synchronises; diagnostics)"""
        st = self.__dict__.get("_amp_state")
        return int(st[2].item()) if st is not None else 0

    def _operands(self) -> str:
        return "fp16" if self.precision == "fp16" else "bf16"

    def _lib(self):
        return L.lib(self._operands())

    # ------------------------------------------------------------------ frozen weights -> HBM layout
    def _backbone_params(self, model):
        return [p for n, p in model.named_parameters() if not n.startswith("CP_") and not n.startswith("head.")]

    def _signature(self, model):
        # (the operand type is part of it: the 16-bit images of the frozen weights are written by the library in use)
        return (self._operands(),) + tuple((p.data_ptr(), p._version) for p in self._backbone_params(model))

    def _ingest(self, model, dev):
        with torch.cuda.device(dev):
            self._ingest_on(model, dev)

    def _ingest_on(self, model, dev):
        """One-time (and after any in-place change / load_state_dict): frozen fp32 parameters ->
        bf16 [depth, out, in] stacks plus transposed copies for the dX GEMMs, fp32 vectors stacked."""
        lib = self._lib()
        stream = lambda: L.stream(dev)   # noqa: E731  (everything below enqueues on dev's current stream)
        blocks = list(model.blocks)
        depth, D = len(blocks), model.embed_dim

        def f32(ts):
            return torch.stack([t.detach().to(dev, torch.float32) for t in ts]).contiguous()

        def bf16_pair(ws, out_f, in_f):
            """-> W, W^T row-major and both once more as K-panel-major images (the layout the GEMMs stage from)."""
            src = f32(ws)  # [depth, out, in]
            w = torch.empty(depth, out_f, in_f, dtype=torch.bfloat16, device=dev)
            wt = torch.empty(depth, in_f, out_f, dtype=torch.bfloat16, device=dev)
            wp, wtp = torch.empty_like(w), torch.empty_like(wt)
            check(lib.cara_f32_to_bf16(ptr(src), ptr(w), C.c_size_t(src.numel()), stream()), "cara_f32_to_bf16")
            for l in range(depth):
                check(lib.cara_transpose_bf16(ptr(w[l]), ptr(wt[l]), out_f, in_f, stream()), "cara_transpose_bf16")
                check(lib.cara_pack_b_panels(ptr(w[l]), in_f, out_f, in_f, ptr(wp[l]), stream()), "cara_pack_b_panels")
                check(lib.cara_pack_b_panels(ptr(wt[l]), out_f, in_f, out_f, ptr(wtp[l]), stream()), "cara_pack_b_panels")
            return w, wt, wp, wtp

        t = {}
        pw = model.patch_embed.proj.weight.detach().to(dev, torch.float32).reshape(D, -1).contiguous()
        t["patch_w"] = torch.empty_like(pw, dtype=torch.bfloat16)
        check(lib.cara_f32_to_bf16(ptr(pw), ptr(t["patch_w"]), C.c_size_t(pw.numel()), stream()), "cara_f32_to_bf16")
        t["patch_b"] = model.patch_embed.proj.bias.detach().to(dev, torch.float32).contiguous()
        t["cls"] = model.cls_token.detach().to(dev, torch.float32).reshape(-1).contiguous()
        t["pos"] = model.pos_embed.detach().to(dev, torch.float32).reshape(-1, D).contiguous()
        t["ln1_g"], t["ln1_b"] = f32([b.norm1.weight for b in blocks]), f32([b.norm1.bias for b in blocks])
        t["ln2_g"], t["ln2_b"] = f32([b.norm2.weight for b in blocks]), f32([b.norm2.bias for b in blocks])
        t["qkv_w"], t["qkv_wt"], t["qkv_wp"], t["qkv_wtp"] = bf16_pair([b.attn.qkv.weight for b in blocks], 3 * D, D)
        t["proj_w"], t["proj_wt"], t["proj_wp"], t["proj_wtp"] = bf16_pair([b.attn.proj.weight for b in blocks], D, D)
        t["fc1_w"], t["fc1_wt"], t["fc1_wp"], t["fc1_wtp"] = bf16_pair([b.mlp.fc1.weight for b in blocks], 4 * D, D)
        t["fc2_w"], t["fc2_wt"], t["fc2_wp"], t["fc2_wtp"] = bf16_pair([b.mlp.fc2.weight for b in blocks], D, 4 * D)
        t["qkv_b"] = f32([b.attn.qkv.bias for b in blocks])
        t["proj_b"] = f32([b.attn.proj.bias for b in blocks])
        t["fc1_b"] = f32([b.mlp.fc1.bias for b in blocks])
        t["fc2_b"] = f32([b.mlp.fc2.bias for b in blocks])
        t["norm_g"] = model.norm.weight.detach().to(dev, torch.float32).contiguous()
        t["norm_b"] = model.norm.bias.detach().to(dev, torch.float32).contiguous()
        w = L.VitWeights(**{n: ptr(t[n]) for n, _ in L.VitWeights._fields_})
        self._ingested = (t, w)
        self._ingest_sig = self._signature(model)

    # ------------------------------------------------------------------ per-shape state
    def _state(self, model, B, img, dev):
        if self._ingested is None or self._ingest_sig != self._signature(model) or self._ingested[0]["cls"].device != dev:
            self._ingest(model, dev)
            self._ws.clear()
        ncls = model.head.out_features
        exact = self.weight_dropout == "exact"
        if self.weight_dropout not in ("off", "exact"):
            raise CaraError(f"weight_dropout must be 'off' or 'exact', not {self.weight_dropout!r}")
        if exact and self.cp_length == 2:
            raise CaraError("cp_length 2 (dense QKV deltas) runs with weight_dropout = 'off' only")
        key = (B, img, ncls, str(dev), exact, self._operands(), self._slot)
        st = self._ws.get(key)
        if st is None:
            pe = model.patch_embed
            geom = L.Geom(len(model.blocks), model.embed_dim, model.blocks[0].attn.num_heads, self.rank, self.Rp, self.scale,
                          self.cp_length)
            # (sized with the exact-mode regions when that mode is selected; a call with wd_exact = 0 on the same
            # workspace -- eval -- simply does not touch them)
            patch = pe.proj.kernel_size[0]
            shape = L.VitShape(B, img, patch, pe.proj.in_channels, (img // patch) ** 2 + 1, ncls,
                               float(model.norm.eps), 1 if exact else 0, float(self.weight_dropout_p), 0)
            nbytes = self._lib().cara_vit_workspace_bytes(C.byref(geom), C.byref(shape))
            if nbytes == 0:
                raise CaraError(f"unsupported geometry for the HIP path: {geom.depth=} {geom.dim=} {geom.heads=} "
                                f"{shape.tokens=} (needs head dim 64, tokens <= 608, dim % 256 == 0)")
            if self._slot == 0 and not self.__dict__.get("_keep_ws"):
                self._ws.clear()  # one live workspace: activations of one step (slots > 0: the two-stream step's second half)
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            st = {"geom": geom, "shape": shape, "ws": ws, "logits": torch.empty(B, ncls, device=dev)}
            self._ws[key] = st
        self._last_key = key
        return st

    def _cp_ptrs(self, cp):
        return L.cp_ptrs(self.cp_fields, cp)

    # ------------------------------------------------------------------ forward / backward
    def _run_forward(self, images, droppath, head_w, head_b, cp, need_backward=True):
        model = self._model()
        if images.ndim != 4 or images.shape[2] != images.shape[3]:
            raise CaraError("images must be [B, C, H, H]")
        images = images.contiguous().float()
        dev = images.device
        with torch.cuda.device(dev):   # the library enqueues on the stream it is given and launches on the current device
            return self._run_forward_on(model, images, droppath, head_w, head_b, cp, need_backward, dev)

    def _run_forward_on(self, model, images, droppath, head_w, head_b, cp, need_backward, dev):
        st = self._state(model, images.shape[0], images.shape[2], dev)
        # weight-space dropout is a train-mode thing (nn.Dropout is the identity in eval); the backward of this
        # forward reads the same struct, i.e. the same seed, and regenerates the masks
        use_exact = self.weight_dropout == "exact" and model.training and self.weight_dropout_p > 0
        if model.training and not use_exact and not self._warned_wd:
            self._warned_wd = True
            import warnings
            warnings.warn("cara_amd: train-mode forwards run the FACTORED adapters without the reference's Dropout(0.1) on the "
                          "materialised dW (cara.py:35,57,81,92); set model._cara_engine.weight_dropout = 'exact' (or the "
                          "'weight_dropout' key of cara()'s config) for the reference's train-mode arithmetic", stacklevel=3)
        st["shape"].wd_exact = 1 if use_exact else 0
        st["shape"].wd_p = float(self.weight_dropout_p)
        if use_exact:
            st["shape"].wd_seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self._gen_cpu).item()) \
                if self.weight_dropout_seed is None else int(self.weight_dropout_seed)
        # eval / no_grad: nothing the backward alone reads is kept (cara_vit_shape::inference)
        st["shape"].inference = 0 if need_backward else 1
        cps = self._cp_ptrs([t.detach().contiguous() for t in cp])
        logits = torch.empty_like(st["logits"])
        check(self._lib().cara_vit_forward(C.byref(st["geom"]), C.byref(st["shape"]), C.byref(self._ingested[1]), C.byref(cps),
                                       ptr(head_w.detach().contiguous()), ptr(head_b.detach().contiguous()), ptr(images),
                                       ptr(droppath), ptr(st["ws"]), ptr(logits), stream(dev)), "cara_vit_forward")
        self._fwd_serial += 1
        self._bwd_ready = self._fwd_serial if need_backward else -1
        return logits

    def _grad_buffers(self, model, dev):
        names = [(n, getattr(model, "CP_" + n)) for n in self.cp_fields] + [("head_w", model.head.weight), ("head_b", model.head.bias)]
        sizes = [p.numel() for _, p in names]
        if self._flat_grad is None or self._flat_grad.numel() != sum(sizes) + 1 or self._flat_grad.device != dev:
            from .dist import flat_views
            # (one word behind the gradients: "a non-finite gradient was written" -- all-reduced with them, so every rank sees it)
            self._flat_grad, self._grad_views = flat_views([(n, p.shape) for n, p in names] + [("_found_inf", (1,))], dev)
        return self._grad_views

    def _run_backward(self, dlogits, droppath, head_w, cp, prescaled=False):
        model = self._model()
        st = self._ws[self._last_key]
        dev = dlogits.device
        with torch.cuda.device(dev):
            return self._run_backward_on(model, st, dlogits, droppath, head_w, cp, dev, prescaled)

    def _run_backward_on(self, model, st, dlogits, droppath, head_w, cp, dev, prescaled=False):
        g = self._grad_buffers(model, dev)
        cps = self._cp_ptrs([t.detach().contiguous() for t in cp])
        gps = L.cp_ptrs(self.cp_fields, [g[n] for n in self.cp_fields])
        dl = dlogits.contiguous().float()
        if self.precision == "fp16":
            # every 16-bit gradient of the pass is loss-scale times larger; the kernels that write the final fp32 gradients divide
            # the scale out again (cara_vit_shape::loss_scale) and raise found_inf on a non-finite value
            amp = self._amp(dev)
            if not prescaled:   # (train_step's cross-entropy has scaled dlogits and cleared the word already)
                dl = dl * amp[0]
                g["_found_inf"].zero_()
            st["shape"].loss_scale, st["shape"].found_inf = ptr(amp), ptr(g["_found_inf"])
        else:
            st["shape"].loss_scale, st["shape"].found_inf = None, None
        check(self._lib().cara_vit_backward(C.byref(st["geom"]), C.byref(st["shape"]), C.byref(self._ingested[1]), C.byref(cps),
                                            ptr(head_w.detach().contiguous()), ptr(dl), ptr(droppath),
                                            ptr(st["ws"]), C.byref(gps), ptr(g["head_w"]), ptr(g["head_b"]), stream(dev)),
              "cara_vit_backward")
        self._bwd_ready = -1
        return g

    def seed_rank_streams(self, seed: int, rank: int = 0) -> None:
        """Give this replica its own DropPath / weight-dropout random streams: generator seed = seed * 2^20 + rank.
        Parameters are not touched (replicas stay identical); only the per-step masks differ between ranks."""
        self._gen_seed = int(seed) * (1 << 20) + int(rank)
        self._gen_dev = None
        self._dp_buf = None      # (masks drawn ahead from the previous stream are dropped)
        self._gen_cpu = torch.Generator().manual_seed(self._gen_seed)

    @staticmethod
    def _norm_device(dev):
        """torch.device with an explicit index ('cuda' / torch.device('cuda') -> the current device): what generators
        and tensors report, so that comparisons against it hold."""
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def _device_generator(self, dev):
        if self._gen_seed is None:
            return None
        dev = self._norm_device(dev)
        if self._gen_dev is None or self._gen_dev.device != dev:
            self._gen_dev = torch.Generator(device=dev).manual_seed(self._gen_seed)
        return self._gen_dev

    def draw_droppath(self, model, B, dev) -> Optional[torch.Tensor]:
        """Per-sample multipliers mask/keep_prob of timm DropPath for both branches of every block,
        [depth, 2, B]; None in eval mode or when every rate is 0."""
        if not model.training:
            return None
        # timm 0.4.12: one `drop_path` per block; newer timm: `drop_path1` / `drop_path2` (same rate for both branches)
        def rate(b):
            r = [float(getattr(getattr(b, n, None), "drop_prob", 0.0) or 0.0) for n in ("drop_path", "drop_path1", "drop_path2")]
            if r[1] != r[2]:
                raise CaraError("drop_path1 and drop_path2 of a block must have the same rate")
            return r[0] if hasattr(b, "drop_path") else r[1]
        rates = [rate(b) for b in model.blocks]
        if not any(r > 0 for r in rates):
            return None
        dev = self._norm_device(dev)
        # the keep probabilities stay on the device: a host -> device copy here would make the host wait for the
        # previous step's kernels at the top of every step (and leave the GPU idle until the queue refills)
        key = (tuple(rates), str(dev))
        if self.__dict__.get("_keep_key") != key:
            self._keep = 1.0 - torch.tensor(rates, device=dev).reshape(-1, 1, 1)
            self._keep_key = key
        keep = self._keep
        gen = self._device_generator(dev)
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            # (inside a hipGraph capture the draw must be a node of the graph; on the CPU there is nothing to amortise)
            return ((keep + torch.rand(len(rates), 2, B, device=dev, generator=gen)).floor_() / keep).contiguous()
        # DROPPATH_AHEAD steps' masks per draw: the four small torch launches of a draw (rand, add, floor, div: ~5 us each, serial
        # between the optimiser and the next forward) are paid once per DROPPATH_AHEAD steps instead of every step (r05).  Same
        # generator, same distribution; a step takes the next [depth, 2, B] slice (contiguous: no kernel).
        bkey = (key, B, id(gen))
        buf = self.__dict__.get("_dp_buf")
        if buf is None or self._dp_key != bkey or self._dp_next >= buf.shape[0]:
            buf = (keep + torch.rand(self.DROPPATH_AHEAD, len(rates), 2, B, device=dev, generator=gen)).floor_() / keep
            self._dp_buf, self._dp_key, self._dp_next = buf, bkey, 0
        out = buf[self._dp_next]
        self._dp_next += 1
        return out

    def forward(self, images, droppath: Optional[torch.Tensor] = None):
        model = self._model()
        if not images.is_cuda:
            raise CaraError("cara_amd runs on the GPU only: move the model and the images to a ROCm device "
                            "(there is no CPU fallback)")
        if not hasattr(model.head, "weight"):
            raise CaraError("the classifier head must be a Linear (num_classes > 0)")
        if droppath is None:
            droppath = self.draw_droppath(model, images.shape[0], images.device)
        cp = [getattr(model, "CP_" + n) for n in self.cp_fields]
        if model.head.weight.device != images.device or cp[0].device != images.device:
            raise CaraError("model parameters and images must be on the same device")
        params = [model.head.weight, model.head.bias, *cp]
        self._need_backward = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if self.precision not in ("bf16", "fp16"):
            raise CaraError(f"precision must be 'bf16' or 'fp16', not {self.precision!r}")
        if self.precision == "fp16" and (self.weight_dropout == "exact" or self.cp_length == 2):
            raise CaraError("precision = 'fp16' runs the factored adapters (weight_dropout = 'off', cp_length 3 / 4 / 5)")
        return _VitFn.apply(self, images, droppath, *params)

    # ------------------------------------------------------------------ fused train step
    def trainable_parameters(self):
        """The selection rule of vit_cp.py:175-183: names containing "CP" or "head"."""
        model = self._model()
        return [getattr(model, "CP_" + n) for n in self.cp_fields] + [model.head.weight, model.head.bias]

    def _apply_gradients(self, optimizer=None, group=None, prescaled=False):
        """Tail of a train step, shared by train_step and the CPU multi-process tests: bind ``p.grad`` of every
        trainable parameter to its view of the flat buffer, ONE all-reduce of that buffer when the group has more than
        one rank -- a plain SUM when the gradients were computed from a dlogits already divided by the world size
        (``prescaled``: train_step), the mean otherwise -- and ``optimizer.step()``."""
        model = self._model()
        g = self._grad_views
        cp = [getattr(model, "CP_" + n) for n in self.cp_fields]
        for n, p in zip(list(self.cp_fields) + ["head_w", "head_b"], cp + [model.head.weight, model.head.bias]):
            if p.grad is None or p.grad.data_ptr() != g[n].data_ptr():
                p.grad = g[n]
        # the only data-path collective of a step: one RCCL all-reduce over xGMI of the flat buffer
        from .dist import allreduce_mean_, allreduce_sum_
        (allreduce_sum_ if prescaled else allreduce_mean_)(self._flat_grad, group)
        if self.precision == "fp16" and self._flat_grad.is_cuda:
            found = g["_found_inf"]
            dev = found.device
            with torch.cuda.device(dev):
                check(self._lib().cara_amp_update(ptr(self._amp(dev)), ptr(found), C.c_float(self.AMP_GROWTH), C.c_float(self.AMP_BACKOFF),
                                                  int(self.AMP_GROWTH_INTERVAL), C.c_float(self.AMP_MAX_SCALE), stream(dev)), "cara_amp_update")
            if optimizer is not None:
                from .optim import AdamW
                if isinstance(optimizer, AdamW):
                    optimizer.skip_flag = found          # the launch changes nothing when the word is set
                    optimizer.step()
                elif float(found.item()) == 0.0:         # a foreign optimiser: the host looks at the word (one sync per step)
                    optimizer.step()
            return
        if optimizer is not None:
            optimizer.step()

    def train_step(self, images, labels, optimizer=None, group=None, droppath: Optional[torch.Tensor] = None):
        """One fine-tuning step of vit_cp.py:45-50 without autograd bookkeeping:
        forward -> mean cross-entropy -> backward straight into ONE flat fp32 gradient buffer
        (12 CP tensors + head) -> a single all-reduce of that buffer when ``group``/the default
        process group has more than one rank -> ``optimizer.step()``.  Returns the loss (device
        scalar, local to this rank).  ``p.grad`` of every trainable parameter is a view of the flat
        buffer, so any torch optimizer consumes it unchanged."""
        model = self._model()
        if not images.is_cuda:
            raise CaraError("cara_amd runs on the GPU only (no CPU fallback)")
        dev = images.device
        if labels.dtype != torch.int64 or labels.device != dev or labels.ndim != 1 or labels.shape[0] != images.shape[0]:
            raise CaraError("labels must be an int64 [batch] tensor on the images' device (the kernel reads 8-byte class indices)")
        cp = [getattr(model, "CP_" + n) for n in self.cp_fields]
        if not hasattr(model.head, "weight"):
            raise CaraError("the classifier head must be a Linear (num_classes > 0)")
        hw, hb = model.head.weight, model.head.bias
        if hw.device != dev or any(t.device != dev for t in cp):
            raise CaraError("model parameters and images must be on the same device")
        if self.precision == "fp16" and (self.weight_dropout == "exact" or self.cp_length == 2):
            raise CaraError("precision = 'fp16' runs the factored adapters (weight_dropout = 'off', cp_length 3 / 4 / 5)")
        with torch.no_grad(), torch.cuda.device(dev):
            if droppath is None:
                droppath = self.draw_droppath(model, images.shape[0], dev)
            logits = self._run_forward(images, droppath, hw, hb, cp)
            B, ncls = logits.shape
            if self.__dict__.get("_loss_buf") is None or self._loss_buf.numel() != 1 + B or self._loss_buf.device != dev:
                self._loss_buf = torch.empty(1 + B, device=dev)
                self._dlogits = torch.empty(B, ncls, device=dev)
            if self._dlogits.shape != logits.shape:
                self._dlogits = torch.empty(B, ncls, device=dev)
            # dlogits leaves the cross-entropy already divided by the world size (the all-reduce is then a plain SUM) and, in the
            # IEEE-half build, multiplied by the loss scale; the same launch clears the step's found-inf word
            from .dist import world_size
            fp16 = self.precision == "fp16"
            gv = self._grad_buffers(model, dev)
            check(self._lib().cara_cross_entropy_ex(ptr(logits), ptr(labels.contiguous()), ptr(self._loss_buf), ptr(self._dlogits),
                                                    B, ncls, C.c_float(1.0 / world_size(group)), ptr(self._amp(dev)) if fp16 else None,
                                                    ptr(gv["_found_inf"]) if fp16 else None, stream(dev)), "cara_cross_entropy_ex")
            self._run_backward(self._dlogits, droppath, hw, cp, prescaled=True)
            self._apply_gradients(optimizer, group, prescaled=True)
        return self._loss_buf[0]

    # module-level entries (cara.cp_attn / cara.cp_mlp): the reference's patched forwards
    def _weights(self, model, dev):
        if self._ingested is None or self._ingest_sig != self._signature(model) or self._ingested[0]["cls"].device != dev:
            self._ingest(model, dev)
            self._ws.clear()
        return self._ingested

    def _module_args(self, x):
        model = self._model()
        if not x.is_cuda:
            raise CaraError("cara_amd runs on the GPU only (no CPU fallback)")
        if x.ndim != 3 or x.shape[2] != model.embed_dim or x.shape[1] > 608:
            raise CaraError("module-level forward expects x of shape [B, N <= 608, embed_dim]")
        if self.cp_length == 2:
            raise CaraError("with cp_length 2 (dense QKV deltas) call the whole model: the module-level Attention.forward / "
                            "Mlp.forward entries run the factored adapters only")
        if self.weight_dropout == "exact" and model.training:
            raise CaraError("the module-level forwards (Attention.forward / Mlp.forward called on their own) run the factored "
                            "adapters only: with weight_dropout = 'exact' in train mode call the whole model, or switch to eval()")
        return model, [getattr(model, "CP_" + n) for n in self.cp_fields]

    def attn_forward(self, child, x):
        from .modules import AttnFn
        model, cp = self._module_args(x)
        with torch.cuda.device(x.device):
            return AttnFn.apply(self, child, x, *cp)

    def mlp_forward(self, child, x):
        from .modules import MlpFn
        model, cp = self._module_args(x)
        with torch.cuda.device(x.device):
            return MlpFn.apply(self, child, x, *cp)
