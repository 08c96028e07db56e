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


def test_jax_key_mapping_against_the_flax_layer_definitions():
    """The mapping checked against what the Google/Flax layers COMPUTE, not against its own inverse: a hand-built dict in
    the published ViT-B_16.npz layout (conv kernel HWIO, per-head query/key/value kernels [D,H,hd] and biases [H,hd],
    out kernel [H,hd,D], Dense kernels [in,out]) is evaluated directly from those definitions with einsums, and the
    same numbers are pushed through jax_to_state_dict -> the timm-0.4.12-shaped eager modules of the oracle."""
    from oracle import cara_oracle as O
    g = torch.Generator().manual_seed(0)
    D, H, hd, P = 768, 12, 64, 16
    rn = lambda *s: (torch.randn(*s, generator=g, dtype=torch.float64) * 0.05).numpy()  # noqa: E731
    pre = "Transformer/encoderblock_0/"
    mha = pre + "MultiHeadDotProductAttention_1/"
    w = {"embedding/kernel": rn(P, P, 3, D), "embedding/bias": rn(D), "cls": rn(1, 1, D),
         "Transformer/posembed_input/pos_embedding": rn(1, 5, D),
         "Transformer/encoder_norm/scale": rn(D), "Transformer/encoder_norm/bias": rn(D),
         "head/kernel": rn(D, 7), "head/bias": rn(7),
         pre + "LayerNorm_0/scale": rn(D), pre + "LayerNorm_0/bias": rn(D), pre + "LayerNorm_2/scale": rn(D), pre + "LayerNorm_2/bias": rn(D),
         pre + "MlpBlock_3/Dense_0/kernel": rn(D, 4 * D), pre + "MlpBlock_3/Dense_0/bias": rn(4 * D),
         pre + "MlpBlock_3/Dense_1/kernel": rn(4 * D, D), pre + "MlpBlock_3/Dense_1/bias": rn(D),
         mha + "out/kernel": rn(H, hd, D), mha + "out/bias": rn(D)}
    for n in ("query", "key", "value"):
        w[mha + n + "/kernel"], w[mha + n + "/bias"] = rn(D, H, hd), rn(H, hd)
    T = lambda k: torch.from_numpy(w[k])  # noqa: E731
    vit = O.create_vit("vit_base_patch16_224_in21k", depth=1, img_size=32, num_classes=7).double()
    sd = jax_to_state_dict(w, vit)
    missing, unexpected = vit.load_state_dict({k: v.double() for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected
    # (1) patch embedding: Flax Conv on NHWC with an HWIO kernel, stride = kernel
    img = torch.randn(2, 3, 32, 32, generator=g, dtype=torch.float64)
    nhwc = img.permute(0, 2, 3, 1).reshape(2, 2, P, 2, P, 3)                       # [b, i, di, j, dj, c]
    flax = torch.einsum("bidjec,deco->bijo", nhwc, T("embedding/kernel")) + T("embedding/bias")
    assert torch.allclose(vit.patch_embed(img), flax.reshape(2, 4, D), atol=1e-10)
    # (2) MultiHeadDotProductAttention: DenseGeneral kernels per head
    x = torch.randn(2, 5, D, generator=g, dtype=torch.float64)
    q, k, v = (torch.einsum("bne,ehd->bhnd", x, T(mha + n + "/kernel")) + T(mha + n + "/bias")[None, :, None, :] for n in ("query", "key", "value"))
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ v               # [b, h, n, d]
    flax = torch.einsum("bhnd,hdo->bno", a, T(mha + "out/kernel")) + T(mha + "out/bias")
    assert torch.allclose(vit.blocks[0].attn(x), flax, atol=1e-9)
    # (3) MlpBlock: Dense kernels are [in, out]
    flax = torch.nn.functional.gelu(x @ T(pre + "MlpBlock_3/Dense_0/kernel") + T(pre + "MlpBlock_3/Dense_0/bias")) \
        @ T(pre + "MlpBlock_3/Dense_1/kernel") + T(pre + "MlpBlock_3/Dense_1/bias")
    assert torch.allclose(vit.blocks[0].mlp(x), flax, atol=1e-9)
    # (4) LayerNorms, cls / position embedding, head
    assert torch.equal(vit.blocks[0].norm1.weight, T(pre + "LayerNorm_0/scale")) and torch.equal(vit.blocks[0].norm2.bias, T(pre + "LayerNorm_2/bias"))
    assert torch.equal(vit.norm.weight, T("Transformer/encoder_norm/scale")) and torch.equal(vit.cls_token, T("cls"))
    assert torch.equal(vit.pos_embed, T("Transformer/posembed_input/pos_embedding"))
    assert torch.allclose(vit.head(x[:, 0]), x[:, 0] @ T("head/kernel") + T("head/bias"), atol=1e-10)


def test_save_best_checkpoint_has_the_reference_key_set(tmp_path):
    """vit_cp.py:61-66,168-173: the best-accuracy file is the WHOLE state dict (timm backbone keys + 12 CP_* + head)
    and loads strictly into a freshly adapted model (--evaluate)."""
    from cara_amd import cara
    from cara_amd.recipe import checkpoint_name, load_checkpoint, save_checkpoint
    mk = lambda seed: (torch.manual_seed(seed), cara({"model": create_model("vit_base_patch16_224_in21k", depth=2, num_classes=21843),  # noqa: E731
                                                        "rank": 4, "scale": 1.0, "l_mu": 1.5, "l_std": 0.1}))[1]
    a = mk(1)
    a.reset_classifier(5)
    path = checkpoint_name("cifar", 0.123456789, 42, str(tmp_path))
    assert path.endswith("vit_cifar_0.12346_seed_42.pt")
    save_checkpoint(a, path)
    sd = torch.load(path)
    keys = set(sd)
    assert {"CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2", "CP_bias1", "CP_bias2", "CP_bias3"} <= keys
    assert {"cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.1.attn.qkv.weight", "blocks.0.mlp.fc2.bias", "norm.weight",
            "head.weight", "head.bias"} <= keys and len(keys) == 12 + 4 + 2 + 2 + 2 * 12
    b = mk(2)
    b.reset_classifier(5)
    load_checkpoint(b, path)
    for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), n
