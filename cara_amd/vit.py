"""Host-side ViT container with timm 0.4.12's module/attribute names and state-dict layout.

timm is not a dependency of this build (and is not installed here); the reference creates its
backbone with ``timm.models.create_model`` (``/root/reference/image_classification/vit_cp.py:155``,
``tests/test_cara.py:19``) and ``src/cara/cara.py:110,147,157`` dispatches on timm's exact
classes.  These classes hold the same parameters under the same names
(``blocks.{i}.attn.qkv.weight`` ...), so checkpoints and ``cara()`` bookkeeping are
interchangeable.  They are parameter containers: all arithmetic runs in libcara_hip.so
(``cara_amd.engine``) -- there is no CPU or eager forward in this package.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from ._lib import CaraError


class _NoEagerForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise CaraError(
            f"{type(self).__name__}.forward has no eager implementation in cara_amd: apply cara_amd.cara() "
            "to the VisionTransformer and call the model on a CUDA (ROCm) tensor; the HIP library does the work.")


class DropPath(nn.Module):
    """timm DropPath (per-sample stochastic depth).  Only holds the rate: the engine draws the
    per-sample multipliers for all blocks at once and feeds them to the residual epilogues."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob in (None, 0.0) or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


class Mlp(_NoEagerForward):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(_NoEagerForward):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(nn.Module):
    """Child order norm1, attn, drop_path, norm2, mlp -- it fixes the index walk of cara()."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
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


class PatchEmbed(_NoEagerForward):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0):
        super().__init__()
        if drop_rate != 0.0 or attn_drop_rate != 0.0:
            raise CaraError("cara_amd supports drop_rate = attn_drop_rate = 0 (the reference's configuration)")
        if mlp_ratio != 4.0 or not qkv_bias:
            raise CaraError("cara_amd supports mlp_ratio 4 and qkv_bias=True (ViT-B/L as the reference uses them)")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.in_chans = in_chans
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, drop_rate, attn_drop_rate, dpr[i], nn.GELU, norm_layer)
            for i in range(depth)])
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
        """vit_cp.py:166 calls this AFTER cara(); the engine re-reads ``self.head`` every step."""
        self.num_classes = num_classes
        dev = self.cls_token.device
        self.head = (nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()).to(dev)

    def forward(self, x):
        eng = self.__dict__.get("_cara_engine")
        if eng is None:
            raise CaraError("this VisionTransformer has no adapters installed: call cara_amd.cara({...}) first "
                            "(the plain timm forward is outside the CaRA hot path and is not provided)")
        return eng.forward(x)


_MODELS = {
    "vit_base_patch16_224_in21k": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=21843),
    "vit_base_patch16_224": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=1000),
    # build-own generalisation; the reference hard-codes ViT-B dims
    "vit_large_patch16_384": dict(img_size=384, patch_size=16, embed_dim=1024, depth=24, num_heads=16, num_classes=1000),
}


def create_model(name: str, checkpoint_path: str = "", drop_path_rate: float = 0.0, **kw) -> VisionTransformer:
    """Counterpart of ``timm.models.create_model`` for the names the reference uses."""
    if name not in _MODELS:
        raise CaraError(f"unknown model {name!r}; known: {sorted(_MODELS)}")
    cfg = dict(_MODELS[name])
    cfg.update(kw)
    model = VisionTransformer(drop_path_rate=drop_path_rate, **cfg)
    if checkpoint_path:
        from .checkpoint import load_jax_npz
        load_jax_npz(model, checkpoint_path)
    return model
