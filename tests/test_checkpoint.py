"""CPU tests of the host-side ingest / recipe helpers."""
import math

import numpy as np
import torch

from cara_amd import create_model
from cara_amd.checkpoint import jax_to_state_dict, load_jax_npz, resize_pos_embed, state_dict_to_jax
from cara_amd.recipe import CosineLRScheduler, trainable_parameters


def test_jax_npz_round_trip(tmp_path):
    torch.manual_seed(0)
    src = create_model("vit_base_patch16_224_in21k", depth=2, num_classes=10)
    with torch.no_grad():
        for p in src.parameters():
            p.copy_(torch.randn_like(p))
    w = state_dict_to_jax(src)
    # the Google layout: per-head kernels and HWIO conv
    assert w["Transformer/encoderblock_0/MultiHeadDotProductAttention_1/query/kernel"].shape == (768, 12, 64)
    assert w["Transformer/encoderblock_1/MultiHeadDotProductAttention_1/out/kernel"].shape == (12, 64, 768)
    assert w["embedding/kernel"].shape == (16, 16, 3, 768)
    assert w["Transformer/encoderblock_0/MlpBlock_3/Dense_0/kernel"].shape == (768, 3072)
    path = str(tmp_path / "vit.npz")
    np.savez(path, **w)
    dst = create_model("vit_base_patch16_224_in21k", depth=2, num_classes=10)
    load_jax_npz(dst, path)
    for (n, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), n
    # create_model(checkpoint_path=...) is the reference's call (vit_cp.py:155)
    again = create_model("vit_base_patch16_224_in21k", checkpoint_path=path, depth=2, num_classes=10)
    assert torch.equal(again.blocks[1].mlp.fc2.weight, src.blocks[1].mlp.fc2.weight)


def test_head_skipped_when_class_count_differs_and_pos_embed_resized():
    src = create_model("vit_base_patch16_224_in21k", depth=1, num_classes=10)
    w = state_dict_to_jax(src)
    dst = create_model("vit_base_patch16_224_in21k", depth=1, num_classes=7, img_size=32)   # 2x2 grid
    sd = jax_to_state_dict(w, dst)
    assert "head.weight" not in sd and sd["pos_embed"].shape == (1, 5, 768)
    same = resize_pos_embed(src.pos_embed.detach(), src.pos_embed.detach())
    assert torch.allclose(same, src.pos_embed.detach(), atol=1e-6)


def test_cosine_schedule_matches_reference_recipe():
    """vit_cp.py:187 stepped per batch with the epoch index (vit_cp.py:55-56)."""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-4)
    s = CosineLRScheduler(opt, t_initial=100, warmup_t=10, lr_min=1e-5, warmup_lr_init=1e-6, decay_rate=0.1)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12             # before the first step
    s.step(0)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12
    s.step(5)
    assert abs(opt.param_groups[0]["lr"] - (1e-6 + 5 * (1e-3 - 1e-6) / 10)) < 1e-12
    s.step(10)                                                        # first cosine epoch: t = 10 (no warm-up prefix)
    assert abs(opt.param_groups[0]["lr"] - (1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * 10 / 100)))) < 1e-12
    s.step(50)
    assert abs(opt.param_groups[0]["lr"] - (1e-5 + 0.5 * (1e-3 - 1e-5))) < 1e-9
    s.step(99)
    assert opt.param_groups[0]["lr"] < 1.1e-5 + 1e-6
    s.step(100)                                                       # past the single cycle
    assert abs(opt.param_groups[0]["lr"] - 1e-5 * 0.1) < 1e-12


def test_trainable_filter():
    from cara_amd import cara
    m = cara({"model": create_model("vit_base_patch16_224_in21k", depth=1, num_classes=5), "rank": 4, "scale": 1.0,
              "l_mu": 1.0, "l_std": 0.0})
    ps = trainable_parameters(m)
    assert len(ps) == 14 and all(p.requires_grad for p in ps)
    assert not m.blocks[0].attn.qkv.weight.requires_grad
