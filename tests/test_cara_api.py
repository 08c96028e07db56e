"""CPU tests of the host-side mirror of the reference interface (no compute on CPU):
the reference's own structural tests (/root/reference/tests/test_cara.py:43-90) plus the
bookkeeping the golden vectors pin, and the C-ABI library's exports."""
import os
import re

import numpy as np
import pytest
import torch
import torch as th

from cara_amd import CaraError, cara, create_model
from cara_amd import _lib

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cara_reference_vectors.npz"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _get_vit(**kw):
    return create_model("vit_base_patch16_224_in21k", drop_path_rate=0.1, **kw)


def _cfg(**kw):
    th.manual_seed(0)
    return {"model": _get_vit(**kw), "rank": 32, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0}


def test_vit_without_cara():
    vit = _get_vit(depth=2)
    for n in ("CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2"):
        assert not hasattr(vit, n)


def test_vit_with_cara():
    vit = cara(_cfg(depth=2))
    for n in ("CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2"):
        assert hasattr(vit, n)


def test_cara_zero_init():
    vit = cara(_cfg(depth=2))
    assert th.allclose(vit.CP_A2, th.zeros_like(vit.CP_A2))
    assert th.allclose(vit.CP_P2, th.zeros_like(vit.CP_P2))


def test_cara_lambda_init():
    vit = cara(_cfg(depth=2))
    assert th.allclose(vit.CP_R1, th.ones_like(vit.CP_R1))
    assert th.allclose(vit.CP_R2, th.ones_like(vit.CP_R2))


def test_returns_same_module_and_missing_key():
    cfg = _cfg(depth=1)
    assert cara(cfg) is cfg["model"]
    with pytest.raises(KeyError):
        cara({"model": _get_vit(depth=1), "rank": 8, "scale": 1.0, "l_mu": 1.0})


def test_parameter_names_shapes_and_walk():
    vit = _get_vit()
    th.manual_seed(0)
    m = cara({"model": vit, "rank": 8, "scale": 0.3, "l_mu": 1.5, "l_std": 0.1})
    names = [n for n, _ in m.named_parameters() if n.startswith("CP_")]
    assert names == ["CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2",
                     "CP_bias1", "CP_bias2", "CP_bias3"]
    shapes = {n: tuple(getattr(m, n).shape) for n in names}
    assert shapes == {"CP_A1": (36, 8), "CP_A2": (768, 8), "CP_A3": (12, 8), "CP_A4": (64, 8), "CP_P1": (108, 8),
                      "CP_P2": (768, 8), "CP_P3": (768, 8), "CP_R1": (8,), "CP_R2": (8,), "CP_bias1": (768,),
                      "CP_bias2": (3072,), "CP_bias3": (768,)}
    walk = [[b.attn.idx, b.attn.attn_idx, b.mlp.idx] for b in m.blocks]
    assert walk == G["idx_walk"].tolist()          # recorded from the reference's set_cara
    assert [m.idx, m.attn_idx] == G["final_idx"].tolist() == [108, 36]
    for b in m.blocks:
        assert b.attn.s == 0.3 and b.mlp.s == 0.3 and b.attn.dim == 8 and isinstance(b.attn.dp, th.nn.Dropout)
        assert b.attn.dp.p == 0.1 and b.mlp.dp.p == 0.1
    # trainable-selection rule of vit_cp.py:175-183 finds exactly CP_* and head.*
    sel = [n for n, _ in m.named_parameters() if "CP" in n or "head" in n]
    assert sel == names + ["head.weight", "head.bias"]
    assert sum(getattr(m, n).numel() for n in names) == 2526 * 8 + 4608


@pytest.mark.parametrize("rank", [8, 16, 32, 64])
def test_init_matches_reference_draws(rank):
    """Same initialisers in the same order => the reference's tensors under the same seed."""
    vit = _get_vit()
    th.manual_seed(0)
    m = cara({"model": vit, "rank": rank, "scale": 1.0, "l_mu": 1.5, "l_std": 0.1})
    for n in ("CP_A1", "CP_A3", "CP_A4", "CP_P1", "CP_R1", "CP_R2"):
        assert th.equal(getattr(m, n).detach(), th.from_numpy(G[f"init_r{rank}_{n}"])), n
    assert th.equal(m.CP_P3.detach()[:8], th.from_numpy(G[f"init_r{rank}_CP_P3_rows0_8"]))


@pytest.mark.parametrize("cp_length", [2, 3, 5])
def test_other_orders_of_the_qkv_tensorisation(cp_length):
    """config["cp_length"] = the `--dims` of image_classification/dim_experiment.py: parameter names, shapes, zero
    factor and index walk of its set_CP (:264-297, :330-336)."""
    cfg = _cfg(depth=3)
    cfg["cp_length"] = cp_length
    vit = cara(cfg)
    shapes = {n: tuple(p.shape) for n, p in vit.named_parameters() if n.startswith("CP_A")}
    if cp_length == 5:
        assert shapes == {"CP_A1": (3, 32), "CP_A2": (3, 32), "CP_A3": (768, 32), "CP_A4": (12, 32), "CP_A5": (64, 32)}
        assert th.count_nonzero(vit.CP_A3) == 0 and th.count_nonzero(vit.CP_A2) > 0
        assert [b.attn.attn_idx for b in vit.blocks] == [0, 1, 2] and vit.attn_idx == 3
    elif cp_length == 3:
        assert shapes == {"CP_A1": (9, 32), "CP_A2": (768, 32), "CP_A3": (768, 32)}
        assert th.count_nonzero(vit.CP_A2) == 0
        assert [b.attn.attn_idx for b in vit.blocks] == [0, 3, 6]
    else:
        assert shapes == {"CP_A1": (9, 32), "CP_A2": (768 * 768, 32)}            # dim_experiment.py:293-297
        assert th.count_nonzero(vit.CP_A2) == 0
        assert [b.attn.attn_idx for b in vit.blocks] == [0, 3, 6]
    assert [b.attn.idx for b in vit.blocks] == [0, 9, 18] and vit.idx == 27 and vit.cp_l == cp_length
    assert vit._cara_engine.cp_fields == _lib.cp_fields(cp_length)


def test_order_2_declaration_draws_and_limits():
    """Order 2 (dense QKV deltas): the parameter draws are the reference script's (same initialisers, same order:
    A1 xavier, A2 zeros, then P1, P3, R1, R2 -- dim_experiment.py:293-312, checked through the oracle's restatement,
    itself pinned to the script in tests/test_oracle.py); the exact weight-dropout mode is refused by name."""
    from oracle import cara_oracle as O
    th.manual_seed(11)
    want = O.init_cp_params(4, 1.5, 0.1, dim=256, heads=4, depth=2, cp_length=2)
    model = _get_vit(embed_dim=256, num_heads=4, depth=2, num_classes=5)
    th.manual_seed(11)
    vit = cara({"model": model, "rank": 4, "scale": 2.0, "l_mu": 1.5, "l_std": 0.1, "cp_length": 2})
    for k, v in want.items():
        assert th.equal(getattr(vit, k).detach(), v), k
    assert [n for n, _ in vit.named_parameters() if n.startswith("CP_")] == list(want.keys())
    cfg = _cfg(depth=2)
    cfg.update(cp_length=2, weight_dropout="exact")
    with pytest.raises(CaraError, match="weight_dropout = 'off' only"):
        cara(cfg)


def test_state_dict_roundtrip_and_reset_classifier():
    m = cara(_cfg(depth=2))
    m.reset_classifier(100)
    sd = m.state_dict()
    for k in ("CP_A1", "CP_bias3", "cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.1.attn.qkv.weight",
              "blocks.0.mlp.fc2.bias", "norm.weight", "head.weight"):
        assert k in sd
    assert sd["head.weight"].shape == (100, 768)
    m2 = cara(_cfg(depth=2))
    m2.reset_classifier(100)
    m2.load_state_dict(sd)
    assert th.equal(m2.CP_A1, m.CP_A1)


def test_no_cpu_fallback():
    m = cara(_cfg(depth=1))
    with pytest.raises(CaraError):
        m(th.randn(2, 3, 224, 224))
    with pytest.raises(CaraError):
        _get_vit(depth=1)(th.randn(1, 3, 224, 224))


def test_two_models_do_not_alias():
    """cara.py:185-186 rebinds a module global; this build binds factors per model."""
    a, b = cara(_cfg(depth=1)), cara(_cfg(depth=1))
    assert a.blocks[0].attn._cara_owner() is a and b.blocks[0].attn._cara_owner() is b


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cara_hip.h")).read()
    declared = set(re.findall(r"\b(cara_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.cara_abi_version() == 14 and lib.cara_build_arch() == b"gfx950"


# ---- drop-in on a FOREIGN timm-shaped model (vit_cp.py:13-15,155 builds it with timm.models.create_model) --------
def test_foreign_timm_shaped_vit_is_accepted_by_structure():
    """cara() dispatches on structure, not on this package's classes: the oracle's restatement of timm's
    VisionTransformer / Attention / Mlp (another module, same names and attributes, eager forwards) gets the same 12
    parameters, index walk and rebound forwards as the package's own container; its whole-model forward is rebound
    to the fused path (which refuses CPU tensors: no fallback)."""
    from oracle import cara_oracle as O
    from cara_amd import cara
    from cara_amd._lib import CaraError
    torch.manual_seed(0)
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=3, num_classes=7)
    assert type(vit).__module__.startswith("oracle")
    out = cara({"model": vit, "rank": 8, "scale": 0.5, "l_mu": 1.0, "l_std": 0.0})
    assert out is vit
    names = [n for n, _ in vit.named_parameters() if n.startswith("CP_")]
    assert names == ["CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_P1", "CP_P2", "CP_P3", "CP_R1", "CP_R2", "CP_bias1", "CP_bias2", "CP_bias3"]
    assert vit.CP_A1.shape == (9, 8) and vit.CP_P1.shape == (27, 8) and vit.idx == 27 and vit.attn_idx == 9
    for l, blk in enumerate(vit.blocks):
        assert (blk.attn.idx, blk.attn.attn_idx, blk.mlp.idx) == (9 * l, 3 * l, 9 * l + 1)
        assert blk.attn.s == 0.5 and blk.mlp.dim == 8 and "forward" in blk.attn.__dict__ and "forward" in blk.mlp.__dict__
    assert "forward" in vit.__dict__ and hasattr(vit, "_cara_engine")
    with pytest.raises(CaraError):
        vit(torch.zeros(1, 3, 224, 224))          # CPU tensor: the fused path has no fallback
    vit.reset_classifier(5)                       # vit_cp.py:166, after cara()
    assert vit.head.out_features == 5


def test_unsupported_vit_is_refused_with_a_reason():
    from oracle import cara_oracle as O
    from cara_amd import cara
    from cara_amd._lib import CaraError
    vit = O.create_vit("vit_base_patch16_224_in21k", depth=1, num_classes=3)
    vit.blocks[0].attn.qkv = torch.nn.Linear(768, 2304, bias=False)
    with pytest.raises(CaraError, match="qkv"):
        cara({"model": vit, "rank": 8, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})
    with pytest.raises(CaraError, match="timm-shaped"):
        cara({"model": torch.nn.Linear(3, 3), "rank": 8, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})


def test_rank_streams_give_every_rank_its_own_masks():
    """identical parameters, per-rank random streams.  Two replicas built from the same seed hold equal
    parameters; after seed_rank_streams(seed, rank) they draw different DropPath masks, reproducibly."""
    from cara_amd import cara, create_model

    def replica(rank):
        torch.manual_seed(14)
        m = cara({"model": create_model("vit_base_patch16_224_in21k", depth=4, num_classes=5, drop_path_rate=0.5),
                  "rank": 4, "scale": 0.1, "l_mu": 1.5, "l_std": 0.1}).train()
        m._cara_engine.seed_rank_streams(2024, rank)
        return m
    a, b, a2 = replica(0), replica(1), replica(0)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
    cpu = torch.device("cpu")
    da, db, da2 = (m._cara_engine.draw_droppath(m, 64, cpu) for m in (a, b, a2))
    assert da.shape == (4, 2, 64) and not torch.equal(da, db) and torch.equal(da, da2)
    assert not torch.equal(a._cara_engine.draw_droppath(a, 64, cpu), da)          # the stream advances


def test_weight_dropout_is_an_explicit_choice():
    from cara_amd import cara, create_model
    from cara_amd._lib import CaraError
    mk = lambda **kw: cara({"model": create_model("vit_base_patch16_224_in21k", depth=1, num_classes=3), "rank": 4,  # noqa: E731
                            "scale": 1.0, "l_mu": 1.0, "l_std": 0.0, **kw})
    assert mk()._cara_engine.weight_dropout == "off" and mk(weight_dropout="exact")._cara_engine.weight_dropout == "exact"
    with pytest.raises(CaraError):
        mk(weight_dropout="maybe")


def _newer_timm_style(depth=2, **feature):
    """A stand-in shaped like a newer timm VisionTransformer (1.x): blocks carry ls1 / ls2 and drop_path1 / drop_path2
    instead of drop_path, Attention carries q_norm / k_norm, Mlp drop1 / drop2 / norm, the model fc_norm / norm_pre /
    patch_drop / global_pool / num_prefix_tokens.  With every extra an Identity (or 0) it computes timm 0.4.12's
    function and must be accepted; `feature` switches one of them on."""
    from oracle import cara_oracle as O
    nn = torch.nn
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=depth, num_classes=3)
    vit.fc_norm, vit.norm_pre, vit.patch_drop = nn.Identity(), nn.Identity(), nn.Identity()
    vit.global_pool, vit.num_prefix_tokens, vit.no_embed_class, vit.reg_token = "token", 1, False, None
    for b in vit.blocks:
        b.ls1, b.ls2 = nn.Identity(), nn.Identity()
        b.drop_path1, b.drop_path2 = b.drop_path, b.drop_path
        del b.drop_path
        b.attn.q_norm, b.attn.k_norm = nn.Identity(), nn.Identity()
        b.mlp.drop1, b.mlp.drop2, b.mlp.norm = nn.Dropout(0.0), nn.Dropout(0.0), nn.Identity()
    blk = vit.blocks[0]
    for k, v in feature.items():
        if k in ("ls1", "ls2"):
            setattr(blk, k, nn.Linear(768, 768))
        elif k in ("q_norm", "k_norm"):
            setattr(blk.attn, k, nn.LayerNorm(64))
        elif k == "mlp_norm":
            blk.mlp.norm = nn.LayerNorm(3072)
        elif k in ("drop1", "drop2"):
            setattr(blk.mlp, k, nn.Dropout(0.1))
        elif k in ("fc_norm", "norm_pre"):
            setattr(vit, k, nn.LayerNorm(768))
        elif k == "patch_drop":
            vit.patch_drop = nn.Dropout(0.25)
        else:
            setattr(vit, k, v)
    return vit


def test_newer_timm_shaped_vit_without_extras_is_accepted_and_draws_droppath():
    from cara_amd import cara
    vit = cara({"model": _newer_timm_style(), "rank": 8, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0}).train()
    dp = vit._cara_engine.draw_droppath(vit, 4, torch.device("cpu"))     # rates read from drop_path1 / drop_path2
    assert dp.shape == (2, 2, 4)
    assert torch.all(dp[0] == 1.0) and all(v == 0.0 or abs(v - 1.0 / 0.9) < 1e-6 for v in dp[1].flatten().tolist())


@pytest.mark.parametrize("feature,word", [
    ({"ls1": 1}, "LayerScale"), ({"q_norm": 1}, "qk_norm"), ({"mlp_norm": 1}, "mlp.norm"), ({"drop2": 1}, "drop2"),
    ({"fc_norm": 1}, "fc_norm"), ({"norm_pre": 1}, "norm_pre"), ({"patch_drop": 1}, "patch_drop"),
    ({"global_pool": "avg"}, "global_pool"), ({"num_prefix_tokens": 5}, "prefix"), ({"no_embed_class": True}, "prefix"),
])
def test_newer_timm_features_the_fused_path_does_not_compute_are_refused(feature, word):
    """a newer timm ViT with qk_norm, LayerScale, fc_norm, avg pooling, register tokens ... passed the
    structure check and would have silently computed timm 0.4.12's function instead."""
    from cara_amd import cara
    from cara_amd._lib import CaraError
    with pytest.raises(CaraError, match=word):
        cara({"model": _newer_timm_style(depth=1, **feature), "rank": 8, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})


def test_pos_embed_must_match_the_patch_grid():
    from oracle import cara_oracle as O
    from cara_amd import cara
    from cara_amd._lib import CaraError
    vit = O.create_vit("vit_base_patch16_224_in21k", depth=1, num_classes=3)
    vit.pos_embed = torch.nn.Parameter(torch.zeros(1, 50, 768))
    vit.patch_embed.img_size = (224, 224)
    with pytest.raises(CaraError, match="pos_embed"):
        cara({"model": vit, "rank": 8, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})


def test_device_generator_is_kept_across_equivalent_device_spellings():
    """an unindexed torch.device('cuda') never compared equal to the generator's 'cuda:0', so the private
    stream was rebuilt from its seed on every call and every step drew the SAME DropPath masks."""
    from cara_amd.engine import CaraEngine
    assert CaraEngine._norm_device("cpu") == torch.device("cpu")
    assert CaraEngine._norm_device(torch.device("cuda", 0)) == torch.device("cuda", 0)
    from cara_amd import cara, create_model
    m = cara({"model": create_model("vit_base_patch16_224_in21k", depth=2, num_classes=3, drop_path_rate=0.5), "rank": 4,
              "scale": 1.0, "l_mu": 1.0, "l_std": 0.0}).train()
    m._cara_engine.seed_rank_streams(7, 0)
    a = m._cara_engine.draw_droppath(m, 32, "cpu")           # a string, then a torch.device: one generator, advancing
    b = m._cara_engine.draw_droppath(m, 32, torch.device("cpu"))
    assert m._cara_engine._gen_dev is not None and not torch.equal(a, b)


def test_fit_does_not_replace_the_global_rng_in_a_single_process(monkeypatch):
    """fit() seeded private mask streams with a default seed of 0 in every run; single-process runs now keep
    torch's global generators (what the reference's torch.manual_seed(args.seed) controls) unless a seed is passed."""
    from cara_amd import cara, create_model, recipe
    m = cara({"model": create_model("vit_base_patch16_224_in21k", depth=1, num_classes=3), "rank": 4, "scale": 1.0, "l_mu": 1.0,
              "l_std": 0.0})
    recipe.fit(m, lambda epoch: [], epochs=1)
    assert m._cara_engine._gen_seed is None
    recipe.fit(m, lambda epoch: [], epochs=1, seed=5)
    assert m._cara_engine._gen_seed == 5 * (1 << 20)
