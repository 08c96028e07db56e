"""CPU oracle for the CaRA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``cara_amd``) never does; it fails loudly without its HIP library.

What is restated here, and from where (all citations relative to ``/root/reference``):

* ``cp_to_tensor``        -- tensorly==0.8.1 (pinned ``pyproject.toml:11``, ``uv.lock:1403-1404``;
                             NOT vendored in the reference).  Published definition
                             ``T[i0..in] = sum_r w_r prod_k F_k[i_k, r]``.
* ``VisionTransformer`` & co -- timm==0.4.12 (pinned ``pyproject.toml:12``, ``uv.lock:1425-1426``;
                             NOT vendored).  Restated from its published source:
                             ``vit_base_patch16_224_in21k`` = patch 16, dim 768, depth 12, heads 12,
                             qkv_bias, LayerNorm eps 1e-6, exact-erf GELU, no pre-logits,
                             21843 classes (``tests/test_cara.py:98``).
* ``attn_as_written`` / ``mlp_as_written`` -- ``src/cara/cara.py:15-60`` / ``:63-95``
                             (materialise dW with cp_to_tensor, second dense GEMM, dropout on dW).
* ``install_cara``         -- ``src/cara/cara.py:98-166`` (parameter shapes, init order, idx walk).
* factored forms / gradient identities -- SURVEY.md Appendix A.3 / A.4 (algebraic consequences of
                             the as-written form; checked against autograd of the as-written form
                             in ``tests/test_oracle.py``).

PARITY PINNING STATUS.  The reference's own tests pin only structure (parameter names, zero
init of CP_A2/CP_P2, lambda init, output shape: ``tests/test_cara.py:43-98``) -- no numeric
vector.  Everything that IS in ``/root/reference`` (slice indices, reshape/permute orders,
which factor is on which side, where ``s`` and the CP biases enter, dropout placement) is pinned
by running the reference's real ``src/cara/cara.py`` in the build container over this module's
timm/tensorly restatements and comparing outputs and gradients (``tests/golden/make_golden.py``;
vectors committed under ``tests/golden/``).  tensorly's and timm's own arithmetic is restated
from their published definitions: at THAT boundary parity is unpinned (neither package is
installed anywhere in this environment) except for the derived known-answer test
"zero-init adapters => logits equal the plain ViT's bit for bit".
"""
from __future__ import annotations

import math
from functools import partial
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# tensorly 0.8.1: cp_to_tensor
# ----------------------------------------------------------------------------------------------


def cp_to_tensor(cp, mask=None):
    """``tensorly.cp_to_tensor((weights, factors))`` -- call sites ``src/cara/cara.py:27,52,76,88``.

    T[i0, ..., in] = sum_r w[r] * F0[i0, r] * ... * Fn[in, r], shape (I0, ..., In), row-major.
    tensorly folds ``(F0 * w) @ khatri_rao(F1..Fn).T`` on mode 0; the same contraction is done
    here mode by mode (identical up to fp rounding order).
    """
    weights, factors = cp
    factors = list(factors)
    shape = [f.shape[0] for f in factors]
    rank = factors[0].shape[1]
    if weights is None:
        weights = torch.ones(rank, dtype=factors[0].dtype)
    lead = factors[0] * weights.reshape(1, rank)  # [I0, R]
    kr = factors[1]  # khatri-rao of the remaining modes, [I1*...*In, R]
    for f in factors[2:]:
        kr = (kr.unsqueeze(1) * f.unsqueeze(0)).reshape(-1, rank)
    full = lead @ kr.t()
    return full.reshape(shape)


# ----------------------------------------------------------------------------------------------
# timm 0.4.12: the pieces of vision_transformer.py / layers the hot path touches
# ----------------------------------------------------------------------------------------------


def drop_path(x, drop_prob: float = 0.0, training: bool = False):
    """timm.models.layers.drop.drop_path (0.4.12): per-sample stochastic depth."""
    if drop_prob == 0.0 or not training:
        return x
    keep_prob = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
    random_tensor.floor_()
    return x.div(keep_prob) * random_tensor


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class Mlp(nn.Module):
    """timm.models.layers.mlp.Mlp (0.4.12)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        x = self.fc1(x)
        x = self.act(x)
        x = self.drop(x)
        x = self.fc2(x)
        x = self.drop(x)
        return x


class Attention(nn.Module):
    """timm.models.vision_transformer.Attention (0.4.12)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        attn = self.attn_drop(attn)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        x = self.proj(x)
        x = self.proj_drop(x)
        return x


class Block(nn.Module):
    """timm.models.vision_transformer.Block (0.4.12).  Child order norm1, attn, drop_path,
    norm2, mlp fixes the idx walk of ``src/cara/cara.py:146-166``."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x


class PatchEmbed(nn.Module):
    """timm.models.layers.patch_embed.PatchEmbed (0.4.12)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x).flatten(2).transpose(1, 2)
        return self.norm(x)


class VisionTransformer(nn.Module):
    """timm.models.vision_transformer.VisionTransformer (0.4.12), non-distilled, no pre-logits."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, drop_rate, attn_drop_rate, dpr[i],
                  act_layer=nn.GELU, norm_layer=norm_layer) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    def reset_classifier(self, num_classes, global_pool=""):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward_features(self, x):
        x = self.patch_embed(x)
        cls_token = self.cls_token.expand(x.shape[0], -1, -1)
        x = torch.cat((cls_token, x), dim=1)
        x = self.pos_drop(x + self.pos_embed)
        x = self.blocks(x)
        x = self.norm(x)
        return self.pre_logits(x[:, 0])

    def forward(self, x):
        return self.head(self.forward_features(x))


def create_vit(name: str = "vit_base_patch16_224_in21k", drop_path_rate: float = 0.0, **kw) -> VisionTransformer:
    """Stand-in for ``timm.models.create_model`` (``tests/test_cara.py:19``, ``vit_cp.py:155``)."""
    table = {
        "vit_base_patch16_224_in21k": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=21843),
        "vit_base_patch16_224": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=1000),
    }
    cfg = dict(table[name])
    cfg.update(kw)
    return VisionTransformer(drop_path_rate=drop_path_rate, **cfg)


# ----------------------------------------------------------------------------------------------
# src/cara/cara.py, restated over explicit tensors
# ----------------------------------------------------------------------------------------------

CP_NAMES = ("CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2",
            "CP_bias1", "CP_bias2", "CP_bias3")


def cp_length_of(cp) -> int:
    """Order of the QKV tensorisation a CP dict carries (``image_classification/dim_experiment.py:264-295``):
    5 has a CP_A5, 3 has no CP_A4, 2 has no CP_A3 either, 4 is ``src/cara``'s."""
    return 5 if "CP_A5" in cp else (4 if "CP_A4" in cp else (3 if "CP_A3" in cp else 2))


def cp_shapes(rank: int, dim: int = 768, heads: int = 12, depth: int = 12, cp_length: int = 4) -> Dict[str, Tuple[int, ...]]:
    """Shapes of ``src/cara/cara.py:112-125`` (dim/heads/depth generalised; reference = 768/12/12); for
    ``cp_length`` 3 / 5 those of ``dim_experiment.py:286-288`` / ``:266-270``."""
    if cp_length == 5:
        a = {"CP_A1": (depth, rank), "CP_A2": (3, rank), "CP_A3": (dim, rank), "CP_A4": (heads, rank),
             "CP_A5": (dim // heads, rank)}
    elif cp_length == 3:
        a = {"CP_A1": (3 * depth, rank), "CP_A2": (dim, rank), "CP_A3": (dim, rank)}
    elif cp_length == 2:   # dim_experiment.py:293-297: rank dense dim x dim matrices per projection
        a = {"CP_A1": (3 * depth, rank), "CP_A2": (dim * dim, rank)}
    else:
        a = {"CP_A1": (3 * depth, rank), "CP_A2": (dim, rank), "CP_A3": (heads, rank), "CP_A4": (dim // heads, rank)}
    a.update({"CP_P1": (9 * depth, rank), "CP_P2": (dim, rank),
              "CP_P3": (dim, rank), "CP_R1": (rank,), "CP_R2": (rank,),
              "CP_bias1": (dim,), "CP_bias2": (4 * dim,), "CP_bias3": (dim,)})
    return a


def init_cp_params(rank: int, l_mu: float, l_std: float, dim=768, heads=12, depth=12, cp_length: int = 4) -> Dict[str, torch.Tensor]:
    """Initialisation of ``src/cara/cara.py:127-142``; consumes the global torch RNG in the same
    order (A1, A3, A4, P1, P3, R1, R2; zeros consume nothing).  ``cp_length`` 3 / 5: the initialisers of
    ``dim_experiment.py:289-292`` / ``:271-276``, in that order."""
    p = {k: torch.empty(*s) for k, s in cp_shapes(rank, dim, heads, depth, cp_length).items()}
    nn.init.xavier_normal_(p["CP_A1"])
    if cp_length == 5:
        nn.init.orthogonal_(p["CP_A2"])
        nn.init.zeros_(p["CP_A3"])
        nn.init.orthogonal_(p["CP_A4"])
        nn.init.orthogonal_(p["CP_A5"])
    elif cp_length == 3:
        nn.init.zeros_(p["CP_A2"])
        nn.init.orthogonal_(p["CP_A3"])
    elif cp_length == 2:
        nn.init.zeros_(p["CP_A2"])
    else:
        nn.init.zeros_(p["CP_A2"])
        nn.init.orthogonal_(p["CP_A3"])
        nn.init.orthogonal_(p["CP_A4"])
    nn.init.xavier_normal_(p["CP_P1"])
    nn.init.zeros_(p["CP_P2"])
    nn.init.orthogonal_(p["CP_P3"])
    if l_std != 0.0:
        nn.init.normal_(p["CP_R1"], mean=l_mu, std=l_std)
        nn.init.normal_(p["CP_R2"], mean=l_mu, std=l_std)
    elif l_mu == 1.0 and l_std == 0.0:
        nn.init.ones_(p["CP_R1"])
        nn.init.ones_(p["CP_R2"])
    # else: left uninitialised, exactly as the reference does (cara.py:134-139)
    for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
        nn.init.zeros_(p[k])
    return p


def qkv_adapter_tensor(cp, attn_idx: int) -> torch.Tensor:
    """The materialised QKV adapter [3, in, out] of one block, for every order of tensorisation
    (``dim_experiment.py:188-207``).  ``attn_idx`` is always the order-4 walk index 3 * block; the order-5 form, whose
    own walk advances by 1 per block (``:334``), takes row ``attn_idx // 3`` of its CP_A1."""
    n = cp_length_of(cp)
    if n == 5:   # :189-194
        f1 = cp["CP_A1"][attn_idx // 3:attn_idx // 3 + 1]
        t = cp_to_tensor((cp["CP_R1"], (f1, cp["CP_A2"], cp["CP_A3"], cp["CP_A4"], cp["CP_A5"]))).squeeze(0)
        K, E, H, D = t.shape
        return t.reshape(K, E, H * D)
    f1 = cp["CP_A1"][attn_idx:attn_idx + 3]
    if n == 3:   # :200-202
        return cp_to_tensor((cp["CP_R1"], (f1, cp["CP_A2"], cp["CP_A3"])))
    if n == 2:   # :203-207: [3, dim * dim] -> [3, in, out]
        t = cp_to_tensor((cp["CP_R1"], (f1, cp["CP_A2"])))
        c = int(round(t.shape[1] ** 0.5))
        return t.reshape(3, c, c)
    t = cp_to_tensor((cp["CP_R1"], (f1, cp["CP_A2"], cp["CP_A3"], cp["CP_A4"])))   # src/cara/cara.py:26-34
    K, E, H, D = t.shape
    return t.reshape(K, E, H * D)


def attn_as_written(x, cp, qkv_w, qkv_b, proj_w, proj_b, *, attn_idx: int, idx: int, s: float,
                    num_heads: int, scale: float, dp=None):
    """``cp_attn`` (``src/cara/cara.py:15-60``) over explicit tensors.  ``dp`` is the
    weight-space dropout module (identity when None/eval); a dict {"qkv": f, "proj": f} gives each call site
    (cara.py:35, :57) its own callable, so that a device path can be fed the same masks."""
    dp = dp or (lambda t: t)
    dp_qkv, dp_proj = (dp["qkv"], dp["proj"]) if isinstance(dp, dict) else (dp, dp)
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, qkv_w, qkv_b)
    t = qkv_adapter_tensor(cp, attn_idx)
    delta = torch.einsum("bnd,kde->kbne", x, dp_qkv(t))
    delta = delta.reshape(3, B, N, num_heads, hd).permute(0, 1, 3, 2, 4)
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    qkv = qkv + delta * s
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * scale
    a = a.softmax(dim=-1)
    y = (a @ v).transpose(1, 2).reshape(B, N, C)
    out = F.linear(y, proj_w, proj_b)
    tp = cp_to_tensor((cp["CP_R2"], (cp["CP_P1"][idx:idx + 1], cp["CP_P2"], cp["CP_P3"])))
    tp = tp.reshape(tp.shape[0] * tp.shape[1], tp.shape[2])
    out = out + (y @ dp_proj(tp.t()) + cp["CP_bias1"]) * s
    return out


def mlp_as_written(x, cp, fc1_w, fc1_b, fc2_w, fc2_b, *, idx: int, s: float, dp=None):
    """``cp_mlp`` (``src/cara/cara.py:63-95``) over explicit tensors (``dp``: callable or {"fc1": f, "fc2": f})."""
    dp = dp or (lambda t: t)
    dp_fc1, dp_fc2 = (dp["fc1"], dp["fc2"]) if isinstance(dp, dict) else (dp, dp)
    up = F.linear(x, fc1_w, fc1_b)
    tu = cp_to_tensor((cp["CP_R2"], (cp["CP_P1"][idx:idx + 4], cp["CP_P2"], cp["CP_P3"])))
    a, b, c = tu.shape
    tu = tu.reshape(a * b, c)
    up = up + (x @ dp_fc1(tu.t()) + cp["CP_bias2"]) * s
    h = F.gelu(up)
    down = F.linear(h, fc2_w, fc2_b)
    td = cp_to_tensor((cp["CP_R2"], (cp["CP_P1"][idx + 4:idx + 8], cp["CP_P2"], cp["CP_P3"])))
    td = td.reshape(a * b, c)
    down = down + (h @ dp_fc2(td) + cp["CP_bias3"]) * s
    return down


def block_indices(depth: int = 12) -> List[Tuple[int, int, int]]:
    """(attn.idx, attn.attn_idx, mlp.idx) per block from the walk at ``cara.py:146-166``:
    Attention takes idx then idx+=1, attn_idx+=3; Mlp takes idx then idx+=8."""
    out, idx, aidx = [], 0, 0
    for _ in range(depth):
        a_idx, a_aidx = idx, aidx
        idx += 1
        aidx += 3
        m_idx = idx
        idx += 8
        out.append((a_idx, a_aidx, m_idx))
    return out


def vit_weights(model: nn.Module) -> Dict[str, torch.Tensor]:
    """Frozen backbone tensors by timm-0.4.12 state-dict key."""
    return {k: v.detach() for k, v in model.state_dict().items() if not k.startswith("CP_")}


def vit_cara_forward(images, w: Dict[str, torch.Tensor], cp: Dict[str, torch.Tensor], *, s: float,
                     depth: int = 12, num_heads: int = 12, patch: int = 16, eps: float = 1e-6,
                     drop_path_keep: Optional[torch.Tensor] = None, factored: bool = False,
                     bf16_sim: bool = False, train: Optional[dict] = None,
                     keep_masks=None, keep_p: float = 0.1, sim_dtype: Optional[torch.dtype] = None,
                     sim_only=None, sim_skip=None):
    """Whole adapted forward (timm VisionTransformer.forward with cp_attn/cp_mlp patched in),
    functional form over the state-dict ``w`` and CP tensors ``cp``.

    ``drop_path_keep``: optional [depth, 2, B] tensor of per-sample branch multipliers
    (mask / keep_prob) replacing timm's in-module RNG draw so that a device path can be fed the
    same masks.  ``factored``: use A.3 instead of materialising dW.  ``bf16_sim``: round to
    bf16 at the points the HIP path rounds (GEMM operands, stored activations) while keeping
    fp32 accumulation, to give a like-for-like comparison basis.
    ``train``: ``{"dp": 0.1, "dpr": [rate per block]}`` reproduces train mode as the reference
    runs it -- Dropout(0.1) on each materialised dW (``cara.py:35,57,81,92``) and timm DropPath
    on each branch -- drawing from the global torch RNG in the reference's order (as-written
    form only).  ``keep_masks(layer, name)`` -> bool [out, in] keep mask of linear ``name`` in {"qkv","proj","fc1",
    "fc2"} replaces the RNG draw of the weight-space dropout (probability ``keep_p``) so that the exact mode of the
    device path, which builds its masks from a counter hash (cara_amd/dropout.py), can be checked element for element.
    """
    dpf = (lambda t: F.dropout(t, train["dp"], True)) if train else None

    def masked(layer):
        # orientation of each dropped tensor at its call site (see attn_as_written / mlp_as_written)
        sc = 1.0 / (1.0 - keep_p)
        mq = keep_masks(layer, "qkv").to(torch.float32)
        mq = mq.reshape(3, mq.shape[0] // 3, mq.shape[1]).permute(0, 2, 1) * sc            # [k, e(in), o']
        mp = keep_masks(layer, "proj").to(torch.float32).t() * sc                           # [c(in), j(out)]
        m1 = keep_masks(layer, "fc1").to(torch.float32).t() * sc                            # [c(in), o]
        m2 = keep_masks(layer, "fc2").to(torch.float32).t() * sc                            # [i(in), o]
        return ({"qkv": lambda t: t * mq, "proj": lambda t: t * mp}, {"fc1": lambda t: t * m1, "fc2": lambda t: t * m2})
    # ``sim_dtype``: the operand type of the rounding model (``bf16_sim`` = torch.bfloat16; torch.float16 asks what the SAME
    # rounding points would cost with fp16 MFMA operands -- 10 mantissa bits at the bf16 MFMA rate)
    if bf16_sim and sim_dtype is None:
        sim_dtype = torch.bfloat16
    # ``sim_only`` / ``sim_skip``: sets of rounding CATEGORIES ("images", "weights", "xn", "T", "qkv", "P", "ao", "h", "head") to
    # enable alone / to leave in fp32 -- the one-category-at-a-time error budget of DESIGN.md section 2 (tools/fp16_sim_study.py)
    r = make_rounder(sim_dtype, sim_only, sim_skip)
    B = images.shape[0]
    dim = w["cls_token"].shape[-1]
    hd = dim // num_heads
    scale = hd ** -0.5
    # patch embed as a GEMM over im2col rows (Conv2d k=s=16)
    gh = images.shape[2] // patch
    cols = images.reshape(B, images.shape[1], gh, patch, gh, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gh, -1)
    pw = w["patch_embed.proj.weight"].reshape(dim, -1)
    x = r(cols, "images") @ r(pw, "weights").t() + w["patch_embed.proj.bias"]
    x = torch.cat((w["cls_token"].expand(B, -1, -1), x), dim=1) + w["pos_embed"]
    idxs = block_indices(depth)
    fac = build_factored(cp, s, depth=depth, heads=num_heads) if factored else None
    for l in range(depth):
        p = f"blocks.{l}."
        a_idx, a_aidx, m_idx = idxs[l]
        xn = r(F.layer_norm(x, (dim,), w[p + "norm1.weight"], w[p + "norm1.bias"], eps), "xn")
        if factored:
            y = _attn_factored(xn, w, p, fac[l], num_heads, scale, r)
        else:
            dp_attn, dp_mlp = masked(l) if keep_masks is not None else (dpf, dpf)
            y = attn_as_written(xn, cp, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"],
                                w[p + "attn.proj.weight"], w[p + "attn.proj.bias"],
                                attn_idx=a_aidx, idx=a_idx, s=s, num_heads=num_heads, scale=scale, dp=dp_attn)
        if train:
            y = drop_path(y, train["dpr"][l], True)
        if drop_path_keep is not None:
            y = y * drop_path_keep[l, 0].reshape(B, 1, 1)
        x = x + y
        xn = r(F.layer_norm(x, (dim,), w[p + "norm2.weight"], w[p + "norm2.bias"], eps), "xn")
        if factored:
            y = _mlp_factored(xn, w, p, fac[l], r)
        else:
            y = mlp_as_written(xn, cp, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"],
                               w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], idx=m_idx, s=s, dp=dp_mlp)
        if train:
            y = drop_path(y, train["dpr"][l], True)
        if drop_path_keep is not None:
            y = y * drop_path_keep[l, 1].reshape(B, 1, 1)
        x = x + y
    xc = F.layer_norm(x[:, 0], (dim,), w["norm.weight"], w["norm.bias"], eps)
    # (the device path runs the final LayerNorm + head in fp32 since round 5 -- cara_head_forward -- so the rounding model
    # rounds nothing here; the "head" category remains for the error-budget study only: tools/fp16_sim_study.py)
    if sim_only is not None and "head" in sim_only:
        return F.linear(r(xc, "head"), r(w["head.weight"], "head"), w["head.bias"])
    return F.linear(xc, w["head.weight"], w["head.bias"])


# ----------------------------------------------------------------------------------------------
# Appendix A.3: factored form  delta = ((x @ U) * 1) @ V'^T,  V' = s * g (.) V
# ----------------------------------------------------------------------------------------------


def khatri_rao(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Row (i*J + j) = a[i] * b[j];  [I,R],[J,R] -> [I*J,R]."""
    return (a.unsqueeze(1) * b.unsqueeze(0)).reshape(-1, a.shape[1])


def build_factored(cp: Dict[str, torch.Tensor], s: float, depth: int = 12, heads: int = 12):
    """Per block, per linear: (U [in,R], Vs [out,R], c_s [out] or None) with
    delta_total = (x @ U) @ Vs^T + c_s,  Vs = s * g (.) V,  c_s = s * CP_bias.
    Table of SURVEY.md A.3 (derived from ``src/cara/cara.py:26-34,51-58,72-82,87-93``)."""
    out = []
    idxs = block_indices(depth)
    n = cp_length_of(cp)
    # out factor [dim, R] (row h*hd+d) and in factor [dim, R] of the QKV adapter, per order of tensorisation
    # (order 2 is not low-rank in (in, out): its QKV entry is the dense, scaled delta itself, ("dense", Dm [3 dim, dim]))
    if n != 2:
        kr_a = cp["CP_A3"] if n == 3 else (khatri_rao(cp["CP_A4"], cp["CP_A5"]) if n == 5 else khatri_rao(cp["CP_A3"], cp["CP_A4"]))
        u_qkv = cp["CP_A3"] if n == 5 else cp["CP_A2"]
    for l in range(depth):
        a_idx, a_aidx, m_idx = idxs[l]
        if n == 2:
            t = qkv_adapter_tensor(cp, a_aidx)                                   # [3, in, out]
            qkv = ("dense", s * t.permute(0, 2, 1).reshape(-1, t.shape[1]), None)   # rows k * dim + out, columns in
        else:
            coef = cp["CP_A1"][l:l + 1] * cp["CP_A2"] if n == 5 else cp["CP_A1"][a_aidx:a_aidx + 3]   # [3,R]
            g_qkv = cp["CP_R1"].unsqueeze(0) * coef  # [3,R]
            v_qkv = (g_qkv.unsqueeze(1) * kr_a.unsqueeze(0)).reshape(-1, kr_a.shape[1]) * s  # [3*dim,R]
            qkv = (u_qkv, v_qkv, None)
        proj = (cp["CP_P3"], s * (cp["CP_R2"] * cp["CP_P1"][a_idx]).unsqueeze(0) * cp["CP_P2"], s * cp["CP_bias1"])
        fc1 = (cp["CP_P3"], s * cp["CP_R2"].unsqueeze(0) * khatri_rao(cp["CP_P1"][m_idx:m_idx + 4], cp["CP_P2"]),
               s * cp["CP_bias2"])
        fc2 = (khatri_rao(cp["CP_P1"][m_idx + 4:m_idx + 8], cp["CP_P2"]),
               s * cp["CP_R2"].unsqueeze(0) * cp["CP_P3"], s * cp["CP_bias3"])
        out.append({"qkv": qkv, "proj": proj, "fc1": fc1, "fc2": fc2})
    return out


def make_rounder(sim_dtype, only=None, skip=None):
    """r(t, category): t rounded to ``sim_dtype`` and back -- for every category, for the categories in ``only``, or for all
    but those in ``skip``; the identity without a ``sim_dtype``."""
    if sim_dtype is None:
        return lambda t, cat=None: t
    only = None if only is None else set(only)
    skip = set(skip or ())

    def r(t, cat=None):
        if (only is not None and cat not in only) or cat in skip:
            return t
        return t.to(sim_dtype).to(t.dtype)
    return r


def adapter_linear(x, wgt, bias, fac, r=lambda t, cat=None: t, xcat=None):
    """y = x W^T + b + (x U) Vs^T + c_s with the rank-R term carried as a K-extension
    ([x | T] [W | Vs]^T), T rounded like any other GEMM operand when ``r`` rounds (``xcat``: the rounding category of x)."""
    U, Vs, cs = fac
    if isinstance(U, str):   # ("dense", Dm, None): two products on the same operand, accumulated in fp32 (the device's B3 form)
        return r(x, xcat) @ r(wgt, "weights").t() + r(x, xcat) @ r(Vs, "weights").t() + bias
    t = r(r(x, xcat) @ r(U, "weights"), "T")
    y = r(x, xcat) @ r(wgt, "weights").t() + t @ r(Vs, "weights").t() + bias
    if cs is not None:
        y = y + cs
    return y


def _attn_factored(xn, w, p, fac, num_heads, scale, r):
    B, N, C = xn.shape
    hd = C // num_heads
    qkv = r(adapter_linear(xn, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"], fac["qkv"], r, "xn"), "qkv")
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    # softmax(S) V as the device evaluates it (cara_amd/csrc/attention.hip): P = exp(S - rowmax) is what gets rounded to
    # bf16 as the MFMA operand, the row sum is taken of the unrounded fp32 P, and the division comes last.  With r = identity
    # this is softmax(S) V.
    sc = (q @ k.transpose(-2, -1)) * scale
    pe = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
    y = r(((r(pe, "P") @ v) / pe.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, N, C), "ao")
    return adapter_linear(y, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"], fac["proj"], r, "ao")


def _mlp_factored(xn, w, p, fac, r):
    up = adapter_linear(xn, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], fac["fc1"], r, "xn")
    h = r(F.gelu(up), "h")   # the device path takes GELU of the fp32 accumulator; its bf16 copy of `up` is for backward only
    return adapter_linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], fac["fc2"], r, "h")


# ----------------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md section 8(d) (shared by tests, smoke and bench; deterministic)
# ----------------------------------------------------------------------------------------------


def synthetic_backbone(depth=12, dim=768, heads=12, num_classes=100, img=224, patch=16, seed=2) -> Dict[str, torch.Tensor]:
    """Synthetic frozen weights in timm-0.4.12 key layout: linear/conv weights trunc-normal
    std 0.02, biases 0, LN gamma 1 beta 0, pos_embed N(0, .02), cls 0 (no .npz offline)."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def tn(*shape):
        t = torch.empty(*shape)
        return nn.init.trunc_normal_(t, std=0.02, generator=g)

    w = {"cls_token": torch.zeros(1, 1, dim),
         "pos_embed": torch.randn(1, (img // patch) ** 2 + 1, dim, generator=g) * 0.02,
         "patch_embed.proj.weight": tn(dim, 3, patch, patch), "patch_embed.proj.bias": torch.zeros(dim)}
    for l in range(depth):
        p = f"blocks.{l}."
        w[p + "norm1.weight"], w[p + "norm1.bias"] = torch.ones(dim), torch.zeros(dim)
        w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"] = tn(3 * dim, dim), torch.zeros(3 * dim)
        w[p + "attn.proj.weight"], w[p + "attn.proj.bias"] = tn(dim, dim), torch.zeros(dim)
        w[p + "norm2.weight"], w[p + "norm2.bias"] = torch.ones(dim), torch.zeros(dim)
        w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"] = tn(4 * dim, dim), torch.zeros(4 * dim)
        w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"] = tn(dim, 4 * dim), torch.zeros(dim)
    w["norm.weight"], w["norm.bias"] = torch.ones(dim), torch.zeros(dim)
    w["head.weight"], w["head.bias"] = tn(num_classes, dim), torch.zeros(num_classes)
    return w


def synthetic_cp(rank=16, l_mu=1.5, l_std=0.1, seed=14, seed_nz=3, dim=768, heads=12, depth=12,
                 nonzero_std=0.05, cp_length: int = 4) -> Dict[str, torch.Tensor]:
    """CP init per ``cara.py:127-142`` under ``torch.manual_seed(seed)`` (cifar hyper-params of
    ``vtab_config.py:2-8``), then CP_A2/CP_P2 ~ N(0, nonzero_std) and small biases so the
    adapter term is non-zero.  Restores the caller's global RNG state."""
    state = torch.get_rng_state()
    try:
        torch.manual_seed(seed)
        cp = init_cp_params(rank, l_mu, l_std, dim, heads, depth, cp_length)
    finally:
        torch.set_rng_state(state)
    g = torch.Generator(device="cpu").manual_seed(seed_nz)
    zero_init = "CP_A3" if cp_length == 5 else "CP_A2"    # the factor the reference initialises to zero
    cp[zero_init] = torch.randn(cp[zero_init].shape, generator=g) * nonzero_std
    cp["CP_P2"] = torch.randn(cp["CP_P2"].shape, generator=g) * nonzero_std
    for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
        cp[k] = torch.randn(cp[k].shape, generator=g) * 0.02
    return cp


def synthetic_batch(batch=64, img=224, num_classes=100, seed_x=0, seed_y=1):
    gx = torch.Generator(device="cpu").manual_seed(seed_x)
    gy = torch.Generator(device="cpu").manual_seed(seed_y)
    x = torch.randn(batch, 3, img, img, generator=gx)
    y = torch.randint(0, num_classes, (batch,), generator=gy)
    return x, y


def train_step_as_written(images, labels, w, cp, head, *, s, depth=12, num_heads=12, keep_masks=None, keep_p=0.1,
                          drop_path_keep=None):
    """One fwd+bwd of the reference's as-written algorithm (dense dW + second GEMM, fp32,
    autograd producing dense ddW) -- the ``cpu_baseline`` workload.  Returns (loss, grads)."""
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    hv = {k: v.clone().requires_grad_(True) for k, v in head.items()}
    ww = dict(w)
    ww["head.weight"], ww["head.bias"] = hv["weight"], hv["bias"]
    logits = vit_cara_forward(images, ww, cpv, s=s, depth=depth, num_heads=num_heads, keep_masks=keep_masks, keep_p=keep_p,
                              drop_path_keep=drop_path_keep)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: v.grad for k, v in cpv.items()}
    grads.update({"head." + k: v.grad for k, v in hv.items()})
    return loss.detach(), logits.detach(), grads
