"""Pretrained-weight ingest: Google/JAX ``ViT-B_16.npz`` -> the timm-0.4.12 parameter layout.

The reference loads its backbone with ``create_model(args.model, checkpoint_path="./ViT-B_16.npz")``
(``/root/reference/image_classification/vit_cp.py:155``); timm 0.4.12's ``_load_weights`` does the
key mapping.  timm is not a dependency here, so the mapping is restated from its published source
(query/key/value kernels [D,H,hd] -> one qkv matrix [3D,D]; out kernel [H,hd,D] -> [D,D]; Dense
kernels transposed; conv kernel HWIO -> OIHW; position embedding resized bilinearly when the token
grid differs).  No ``.npz`` is available offline, so the mapping is checked (i) against a hand-built dict in the
Flax layer layout with the forward evaluated from the Flax definitions (``tests/test_checkpoint.py::
test_jax_key_mapping_against_the_flax_layer_definitions`` on the CPU, ``tests/test_model_gpu.py::
test_flax_layout_npz_through_the_device_path`` through ``create_model(checkpoint_path=...)`` -> ``cara()`` -> the
HIP forward) and (ii) by a round trip through the inverse mapping -- not against a real file.

Host-side, one-time plumbing: the engine converts the loaded fp32 parameters to its bf16 HBM
layout on the next forward (``CaraEngine._ingest``).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _n2p(w: np.ndarray, t: bool = True) -> torch.Tensor:
    if w.ndim == 4 and w.shape[0] == w.shape[1] == w.shape[2] == 1:
        w = w.flatten()
    if t:
        if w.ndim == 4:
            w = w.transpose([3, 2, 0, 1])
        elif w.ndim == 3:
            w = w.transpose([2, 0, 1])
        elif w.ndim == 2:
            w = w.transpose([1, 0])
    return torch.from_numpy(np.ascontiguousarray(w))


def resize_pos_embed(posemb: torch.Tensor, posemb_new: torch.Tensor, num_tokens: int = 1) -> torch.Tensor:
    """Bilinear resize of the grid part of a position embedding (timm 0.4.12 ``resize_pos_embed``)."""
    ntok_new = posemb_new.shape[1]
    tok, grid = posemb[:, :num_tokens], posemb[0, num_tokens:]
    ntok_new -= num_tokens
    gs_old, gs_new = int(math.sqrt(len(grid))), int(math.sqrt(ntok_new))
    grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gs_new, gs_new), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, gs_new * gs_new, -1)
    return torch.cat([tok, grid], dim=1)


def jax_to_state_dict(w: Dict[str, np.ndarray], model) -> Dict[str, torch.Tensor]:
    """timm-0.4.12-keyed tensors for every backbone parameter of ``model`` found in the JAX dict."""
    prefix = "opt/target/" if "opt/target/embedding/kernel" in w else ""
    sd: Dict[str, torch.Tensor] = {}
    sd["patch_embed.proj.weight"] = _n2p(w[f"{prefix}embedding/kernel"])
    sd["patch_embed.proj.bias"] = _n2p(w[f"{prefix}embedding/bias"])
    sd["cls_token"] = _n2p(w[f"{prefix}cls"], t=False)
    pos = _n2p(w[f"{prefix}Transformer/posembed_input/pos_embedding"], t=False)
    if pos.shape != model.pos_embed.shape:
        pos = resize_pos_embed(pos, model.pos_embed)
    sd["pos_embed"] = pos
    sd["norm.weight"] = _n2p(w[f"{prefix}Transformer/encoder_norm/scale"])
    sd["norm.bias"] = _n2p(w[f"{prefix}Transformer/encoder_norm/bias"])
    hb = f"{prefix}head/bias"
    if hasattr(model.head, "bias") and hb in w and model.head.bias.shape[0] == w[hb].shape[-1]:
        sd["head.weight"] = _n2p(w[f"{prefix}head/kernel"])
        sd["head.bias"] = _n2p(w[hb])
    for i in range(len(model.blocks)):
        bp = f"{prefix}Transformer/encoderblock_{i}/"
        mha = bp + "MultiHeadDotProductAttention_1/"
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = _n2p(w[bp + "LayerNorm_0/scale"])
        sd[p + "norm1.bias"] = _n2p(w[bp + "LayerNorm_0/bias"])
        sd[p + "attn.qkv.weight"] = torch.cat([_n2p(w[f"{mha}{n}/kernel"], t=False).flatten(1).T for n in ("query", "key", "value")])
        sd[p + "attn.qkv.bias"] = torch.cat([_n2p(w[f"{mha}{n}/bias"], t=False).reshape(-1) for n in ("query", "key", "value")])
        sd[p + "attn.proj.weight"] = _n2p(w[mha + "out/kernel"]).flatten(1)
        sd[p + "attn.proj.bias"] = _n2p(w[mha + "out/bias"])
        for r in range(2):
            sd[p + f"mlp.fc{r + 1}.weight"] = _n2p(w[bp + f"MlpBlock_3/Dense_{r}/kernel"])
            sd[p + f"mlp.fc{r + 1}.bias"] = _n2p(w[bp + f"MlpBlock_3/Dense_{r}/bias"])
        sd[p + "norm2.weight"] = _n2p(w[bp + "LayerNorm_2/scale"])
        sd[p + "norm2.bias"] = _n2p(w[bp + "LayerNorm_2/bias"])
    return sd


def state_dict_to_jax(model) -> Dict[str, np.ndarray]:
    """Inverse mapping (export in the Google layout); used by the round-trip test."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    D = model.embed_dim
    H = model.blocks[0].attn.num_heads
    hd = D // H
    w: Dict[str, np.ndarray] = {}
    w["embedding/kernel"] = sd["patch_embed.proj.weight"].permute(2, 3, 1, 0).numpy()
    w["embedding/bias"] = sd["patch_embed.proj.bias"].numpy()
    w["cls"] = sd["cls_token"].numpy()
    w["Transformer/posembed_input/pos_embedding"] = sd["pos_embed"].numpy()
    w["Transformer/encoder_norm/scale"] = sd["norm.weight"].numpy()
    w["Transformer/encoder_norm/bias"] = sd["norm.bias"].numpy()
    if "head.weight" in sd:
        w["head/kernel"] = sd["head.weight"].t().numpy()
        w["head/bias"] = sd["head.bias"].numpy()
    for i in range(len(model.blocks)):
        bp = f"Transformer/encoderblock_{i}/"
        mha = bp + "MultiHeadDotProductAttention_1/"
        p = f"blocks.{i}."
        w[bp + "LayerNorm_0/scale"], w[bp + "LayerNorm_0/bias"] = sd[p + "norm1.weight"].numpy(), sd[p + "norm1.bias"].numpy()
        w[bp + "LayerNorm_2/scale"], w[bp + "LayerNorm_2/bias"] = sd[p + "norm2.weight"].numpy(), sd[p + "norm2.bias"].numpy()
        qkv_w, qkv_b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):
            w[f"{mha}{n}/kernel"] = qkv_w[j * D:(j + 1) * D].t().reshape(D, H, hd).numpy()
            w[f"{mha}{n}/bias"] = qkv_b[j * D:(j + 1) * D].reshape(H, hd).numpy()
        w[mha + "out/kernel"] = sd[p + "attn.proj.weight"].reshape(D, H, hd).permute(1, 2, 0).numpy()
        w[mha + "out/bias"] = sd[p + "attn.proj.bias"].numpy()
        for r in range(2):
            w[bp + f"MlpBlock_3/Dense_{r}/kernel"] = sd[p + f"mlp.fc{r + 1}.weight"].t().numpy()
            w[bp + f"MlpBlock_3/Dense_{r}/bias"] = sd[p + f"mlp.fc{r + 1}.bias"].numpy()
    return w


def load_jax_npz(model, path: str) -> None:
    """``create_model(..., checkpoint_path=path)`` of the reference (vit_cp.py:155)."""
    w = np.load(path)
    sd = jax_to_state_dict({k: w[k] for k in w.files}, model)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not k.startswith("CP_") and not k.startswith("head.")]
    if bad or unexpected:
        raise RuntimeError(f"checkpoint {path}: missing {bad}, unexpected {unexpected}")
