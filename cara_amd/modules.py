"""Module-level entries: the reference's patched ``Attention.forward`` (``cp_attn``,
``/root/reference/src/cara/cara.py:15-60``) and ``Mlp.forward`` (``cp_mlp``, ``:63-95``) as
autograd Functions over the per-op C ABI (skinny contraction + K-extension GEMM + fused
attention).  This is the compatibility path for code that calls blocks or sub-modules directly;
``VisionTransformer.__call__`` uses the fused whole-model path (``engine.py``) instead.

Gradients flow to the input and to the 12 shared CP tensors (each module contributes its own
layer's share; autograd sums the shares).  LayerNorm, DropPath and the residual adds of
``Block.forward`` stay ordinary torch ops on this path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from ._lib import check, ptr, stream


class FactorPack:
    """bf16 operand pack of ``cara_factor_prep`` for all layers, rebuilt when a CP tensor changes."""

    def __init__(self, engine):
        self.eng = engine
        self.sig = None
        self.pack = None
        self.lay = None
        self.geom = None

    def get(self, model, dev):
        cp = [getattr(model, "CP_" + n) for n in self.eng.cp_fields]
        sig = tuple((p.data_ptr(), p._version) for p in cp) + (str(dev), self.eng._operands())
        if sig != self.sig:
            eng = self.eng
            t, _ = eng._weights(model, dev)
            geom = L.Geom(len(model.blocks), model.embed_dim, model.blocks[0].attn.num_heads, eng.rank, eng.Rp, eng.scale,
                          eng.cp_length)
            lay = L.PackLayout()
            check(L.lib().cara_pack_offsets(C.byref(geom), C.byref(lay)), "cara_pack_offsets")
            if self.pack is None or self.pack.numel() != lay.total or self.pack.device != dev:
                self.pack = torch.zeros(lay.total, dtype=torch.uint8, device=dev)
            cps = L.cp_ptrs(eng.cp_fields, [p.detach().contiguous() for p in cp])
            check(L.lib().cara_factor_prep(C.byref(geom), C.byref(cps), ptr(t["proj_b"]), ptr(t["fc1_b"]), ptr(t["fc2_b"]),
                                           ptr(self.pack), stream()), "cara_factor_prep")
            self.sig, self.lay, self.geom = sig, lay, geom
        return self

    def op(self, layer, name, rows, cols, dtype=None):
        dtype = dtype or L.act_dtype(self.eng._operands())   # (the pack is written by the library of the engine's precision)
        off = layer * self.lay.layer_stride + getattr(self.lay, name)
        n = rows * cols * (4 if dtype == torch.float32 else 2)
        return self.pack[off:off + n].view(dtype).reshape(rows, cols)


def _lin_fwd(x, W, bias, Ut, Vs, out, epi, ldt, **kw):
    """T = x U ; out = [x | T] [W | Vs]^T + bias -> epilogue.  Returns T^T (kept for dVs)."""
    M = x.shape[0]
    Rp = Ut.shape[0]
    T = torch.empty(M, Rp, dtype=L.act_dtype(), device=x.device)
    Tt = torch.zeros(Rp, ldt, dtype=L.act_dtype(), device=x.device)
    L.skinny_xu(x, Ut, T, Tt)
    L.gemm(x, W, out, epi=epi, bias=bias, A2=T, B2=Vs, **kw)
    return Tt


def _lin_bwd(dy, x_saved, Wt, Vst, U, Tt_saved, dx_out, epi, ldt, want_dc, **kw):
    """G' = dY Vs ; dX = [dY | G'] [W^T | U]^T ; dU = X^T G' ; dVs = dY^T T ; dc = colsum dY."""
    M, out_f = dy.shape
    in_f = x_saved.shape[1]
    Rp = Vst.shape[0]
    dev = dy.device
    G = torch.empty(M, Rp, dtype=L.act_dtype(), device=dev)
    Gt = torch.zeros(Rp, ldt, dtype=L.act_dtype(), device=dev)
    L.skinny_xu(dy, Vst, G, Gt)
    L.gemm(dy, Wt, dx_out, epi=epi, A2=G, B2=U, **kw)
    dU = torch.empty(in_f, Rp, device=dev)
    dVs = torch.empty(out_f, Rp, device=dev)
    dc = torch.empty(out_f, device=dev) if want_dc else None
    L.tskinny_xtg(x_saved, Gt, dU)
    L.tskinny_xtg(dy, Tt_saved, dVs, dc)
    return dU, dVs, dc


def _scatter(eng, model, dev, layer, pieces):
    """Per-layer dU/dVs/dc -> this layer's share of the 12 CP gradients (cara_factor_grad_reduce on
    layer buffers that are zero everywhere except ``layer``)."""
    fp = eng._factors.get(model, dev)
    geom, depth, D, Rp = fp.geom, fp.geom.depth, fp.geom.dim, fp.geom.Rp
    bufs = eng.__dict__.setdefault("_layer_grad_bufs", {})
    key = (str(dev), depth, D, Rp)
    if key not in bufs:
        ins = {"qkv": D, "proj": D, "fc1": D, "fc2": 4 * D}
        outs = {"qkv": 3 * D, "proj": D, "fc1": 4 * D, "fc2": D}
        b = {}
        for n in ("qkv", "proj", "fc1", "fc2"):
            b["dU_" + n] = torch.zeros(depth, ins[n], Rp, device=dev)
            b["dVs_" + n] = torch.zeros(depth, outs[n], Rp, device=dev)
            if n != "qkv":
                b["dc_" + n] = torch.zeros(depth, outs[n], device=dev)
        b["scratch"] = torch.empty(L.lib().cara_factor_grad_scratch_bytes(C.byref(geom)), dtype=torch.uint8, device=dev)
        bufs[key] = b
    b = bufs[key]
    for k, v in pieces.items():
        b[k][layer].copy_(v)
    cp = [getattr(model, "CP_" + n).detach().contiguous() for n in eng.cp_fields]
    grads = [torch.empty_like(p) for p in cp]
    lg = L.LayerGrads(*[ptr(b[n]) for n, _ in L.LayerGrads._fields_])
    check(L.lib().cara_factor_grad_reduce(C.byref(geom), C.byref(L.cp_ptrs(eng.cp_fields, cp)), C.byref(lg),
                                          C.byref(L.cp_ptrs(eng.cp_fields, grads)), ptr(b["scratch"]), stream()),
          "cara_factor_grad_reduce")
    for k in pieces:
        b[k][layer].zero_()
    return grads


class AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, child, x, *cp):
        with L.using(eng._operands()):
            return AttnFn._forward(ctx, eng, child, x, *cp)

    @staticmethod
    def backward(ctx, dy):
        with L.using(ctx.eng._operands()):
            return AttnFn._backward(ctx, dy)

    @staticmethod
    def _forward(ctx, eng, child, x, *cp):
        model = eng._model()
        dev = x.device
        t, _ = eng._weights(model, dev)
        fp = eng._factors.get(model, dev)
        l = child.idx // 9                      # attn.idx = 9 * block (cara.py:151,154)
        B, N, Cd = x.shape
        M, H = B * N, child.num_heads
        ldt = (M + 31) // 32 * 32
        xb = x.detach().reshape(M, Cd).to(L.act_dtype()).contiguous()
        qkv = torch.empty(M, 3 * Cd, dtype=L.act_dtype(), device=dev)
        Tt1 = _lin_fwd(xb, t["qkv_w"][l], t["qkv_b"][l], fp.op(l, "Ut_qkv", fp.geom.Rp, Cd), fp.op(l, "Vs_qkv", 3 * Cd, fp.geom.Rp),
                       qkv, L.EPI_BF16, ldt)
        ao = torch.empty(M, Cd, dtype=L.act_dtype(), device=dev)
        lse = torch.empty(B, H, N, device=dev)
        check(L.lib().cara_attention_fwd(ptr(qkv), ptr(ao), ptr(lse), B, N, H, C.c_float(child.scale), stream()), "cara_attention_fwd")
        y = torch.empty(M, Cd, device=dev)
        Tt2 = _lin_fwd(ao, t["proj_w"][l], fp.op(l, "bias_proj", 1, Cd, torch.float32).reshape(-1),
                       fp.op(l, "Ut_proj", fp.geom.Rp, Cd), fp.op(l, "Vs_proj", Cd, fp.geom.Rp), y, L.EPI_F32, ldt)
        ctx.eng, ctx.child, ctx.l, ctx.shape = eng, child, l, (B, N, Cd, H, ldt)
        ctx.save_for_backward(xb, qkv, lse, ao, Tt1, Tt2)
        return y.reshape(B, N, Cd).to(x.dtype)

    @staticmethod
    def _backward(ctx, dy):
        xb, qkv, lse, ao, Tt1, Tt2 = ctx.saved_tensors
        eng, child, l = ctx.eng, ctx.child, ctx.l
        B, N, Cd, H, ldt = ctx.shape
        model, dev, M = eng._model(), dy.device, B * N
        t, _ = eng._weights(model, dev)
        fp = eng._factors.get(model, dev)
        Rp = fp.geom.Rp
        dyb = dy.reshape(M, Cd).to(L.act_dtype()).contiguous()
        dao = torch.empty(M, Cd, dtype=L.act_dtype(), device=dev)
        dU_p, dV_p, dc_p = _lin_bwd(dyb, ao, t["proj_wt"][l], fp.op(l, "Vst_proj", Rp, Cd), fp.op(l, "U_proj", Cd, Rp), Tt2, dao,
                                    L.EPI_BF16, ldt, True)
        dqkv = torch.empty_like(qkv)
        check(L.lib().cara_attention_bwd(ptr(qkv), ptr(ao), ptr(dao), ptr(lse), ptr(dqkv), B, N, H, C.c_float(child.scale), stream()),
              "cara_attention_bwd")
        dx = torch.empty(M, Cd, device=dev)
        dU_q, dV_q, _ = _lin_bwd(dqkv, xb, t["qkv_wt"][l], fp.op(l, "Vst_qkv", Rp, 3 * Cd), fp.op(l, "U_qkv", Cd, Rp), Tt1, dx,
                                 L.EPI_F32, ldt, False)
        grads = _scatter(eng, model, dev, l, {"dU_qkv": dU_q, "dVs_qkv": dV_q, "dU_proj": dU_p, "dVs_proj": dV_p, "dc_proj": dc_p})
        return (None, None, dx.reshape(B, N, Cd).to(dy.dtype), *grads)


class MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, child, x, *cp):
        with L.using(eng._operands()):
            return MlpFn._forward(ctx, eng, child, x, *cp)

    @staticmethod
    def backward(ctx, dy):
        with L.using(ctx.eng._operands()):
            return MlpFn._backward(ctx, dy)

    @staticmethod
    def _forward(ctx, eng, child, x, *cp):
        model = eng._model()
        dev = x.device
        t, _ = eng._weights(model, dev)
        fp = eng._factors.get(model, dev)
        l = (child.idx - 1) // 9                # mlp.idx = 9 * block + 1 (cara.py:161-162)
        B, N, Cd = x.shape
        M, Rp = B * N, fp.geom.Rp
        ldt = (M + 31) // 32 * 32
        xb = x.detach().reshape(M, Cd).to(L.act_dtype()).contiguous()
        h = torch.empty(M, 4 * Cd, dtype=L.act_dtype(), device=dev)
        u = torch.empty_like(h)
        Tt1 = _lin_fwd(xb, t["fc1_w"][l], fp.op(l, "bias_fc1", 1, 4 * Cd, torch.float32).reshape(-1),
                       fp.op(l, "Ut_fc1", Rp, Cd), fp.op(l, "Vs_fc1", 4 * Cd, Rp), h, L.EPI_GELU, ldt, C2=u)
        y = torch.empty(M, Cd, device=dev)
        Tt2 = _lin_fwd(h, t["fc2_w"][l], fp.op(l, "bias_fc2", 1, Cd, torch.float32).reshape(-1),
                       fp.op(l, "Ut_fc2", Rp, 4 * Cd), fp.op(l, "Vs_fc2", Cd, Rp), y, L.EPI_F32, ldt)
        ctx.eng, ctx.l, ctx.shape = eng, l, (B, N, Cd, ldt)
        ctx.save_for_backward(xb, u, h, Tt1, Tt2)
        return y.reshape(B, N, Cd).to(x.dtype)

    @staticmethod
    def _backward(ctx, dy):
        xb, u, h, Tt1, Tt2 = ctx.saved_tensors
        eng, l = ctx.eng, ctx.l
        B, N, Cd, ldt = ctx.shape
        model, dev, M = eng._model(), dy.device, B * N
        t, _ = eng._weights(model, dev)
        fp = eng._factors.get(model, dev)
        Rp = fp.geom.Rp
        dyb = dy.reshape(M, Cd).to(L.act_dtype()).contiguous()
        dh = torch.empty(M, 4 * Cd, dtype=L.act_dtype(), device=dev)
        dU2, dV2, dc2 = _lin_bwd(dyb, h, t["fc2_wt"][l], fp.op(l, "Vst_fc2", Rp, Cd), fp.op(l, "U_fc2", 4 * Cd, Rp), Tt2, dh,
                                 L.EPI_DGELU, ldt, True, aux=u)
        dx = torch.empty(M, Cd, device=dev)
        dU1, dV1, dc1 = _lin_bwd(dh, xb, t["fc1_wt"][l], fp.op(l, "Vst_fc1", Rp, 4 * Cd), fp.op(l, "U_fc1", Cd, Rp), Tt1, dx,
                                 L.EPI_F32, ldt, True)
        grads = _scatter(eng, model, dev, l, {"dU_fc1": dU1, "dVs_fc1": dV1, "dc_fc1": dc1, "dU_fc2": dU2, "dVs_fc2": dV2, "dc_fc2": dc2})
        return (None, None, dx.reshape(B, N, Cd).to(dy.dtype), *grads)
