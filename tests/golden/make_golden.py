"""Generate tests/golden/*.npz by running the REAL reference adapter code.

Runs only in the build container (``/root/reference`` present).  It loads the reference's own
``src/cara/cara.py`` (by file path, so nothing of it is copied) with ``tensorly`` and ``timm``
resolved to the shims under ``tests/golden/_standins`` -- those shims re-export the oracle's
restatement of the two un-vendored third-party packages (see oracle/cara_oracle.py header for
what that does and does not pin).  It then

  1. records what the reference's ``cara()`` / ``set_cara`` / ``cp_attn`` / ``cp_mlp`` produce
     (index walk, init tensors, module outputs, logits, CP gradients) on seeded inputs, and
  2. asserts that the oracle's functional restatement agrees with it (eval mode, train mode
     with the reference's RNG draw order, and the zero-init known-answer test).

Only inputs' seeds and the reference's OUTPUT tensors are stored (data, not source).

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, ROOT)

from oracle import cara_oracle as O  # noqa: E402
from tests.golden.inputs import randomise_cp, seeded_backbone_into  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location("_ref_cara", os.path.join(REF, "src/cara/cara.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_dim_experiment():
    """image_classification/dim_experiment.py, in place: its sibling modules (vtab.py, vtab_config.py) are the
    reference's own files and resolve from its directory."""
    d = os.path.join(REF, "image_classification")
    sys.path.insert(0, d)
    try:
        spec = importlib.util.spec_from_file_location("_ref_dim_experiment", os.path.join(d, "dim_experiment.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
    return mod


def cp_of(model):
    return {n: getattr(model, n).detach().clone() for n in O.CP_NAMES}


def main():
    ref = load_reference()
    out = {}
    torch.set_num_threads(8)

    # ---- 1. index walk + init tensors (reference tests/test_cara.py config: rank 32, seed 0) ----
    for rank in (8, 16, 32, 64):
        vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=12, num_classes=10)
        torch.manual_seed(0)
        m = ref.cara({"model": vit, "rank": rank, "scale": 1.0, "l_mu": 1.5, "l_std": 0.1})
        assert m is vit
        if rank == 8:
            walk = [(b.attn.idx, b.attn.attn_idx, b.mlp.idx) for b in m.blocks]
            out["idx_walk"] = np.array(walk, dtype=np.int64)
            out["final_idx"] = np.array([m.idx, m.attn_idx], dtype=np.int64)
            assert walk == O.block_indices(12), (walk, O.block_indices(12))
            assert [n for n, _ in m.named_parameters() if n.startswith("CP_")] == list(O.CP_NAMES)
        torch.manual_seed(0)
        mine = O.init_cp_params(rank, 1.5, 0.1)
        for n in O.CP_NAMES:
            assert torch.equal(mine[n], getattr(m, n).detach()), (rank, n)
        # store small ones fully, large ones as a few probes
        for n in ("CP_A1", "CP_A3", "CP_A4", "CP_P1", "CP_R1", "CP_R2"):
            out[f"init_r{rank}_{n}"] = getattr(m, n).detach().numpy().copy()
        out[f"init_r{rank}_CP_P3_rows0_8"] = m.CP_P3.detach()[:8].numpy().copy()
        out[f"init_r{rank}_CP_P3_sum"] = np.array([m.CP_P3.detach().double().sum().item(),
                                                   m.CP_P3.detach().double().abs().sum().item()])
        del vit, m
    # lambda = (1, 0) => ones (tests/test_cara.py:86-90)
    vit = O.create_vit("vit_base_patch16_224_in21k", depth=1, num_classes=10)
    m = ref.cara({"model": vit, "rank": 4, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})
    assert torch.equal(m.CP_R1.detach(), torch.ones(4)) and torch.equal(m.CP_R2.detach(), torch.ones(4))

    # ---- 2. module-level vectors: one Attention and one Mlp in the middle of the walk ----
    R, S = 8, 0.1
    torch.manual_seed(14)
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=12, num_classes=100, img_size=32)
    seeded_backbone_into(vit, 101)
    m = ref.cara({"model": vit, "rank": R, "scale": S, "l_mu": 1.5, "l_std": 0.1})
    randomise_cp(m, 102)
    m.eval()
    cp = cp_of(m)
    w = O.vit_weights(m)
    gx = torch.Generator(device="cpu").manual_seed(103)
    x_mod = torch.randn(2, 7, 768, generator=gx)
    L = 5
    blk = m.blocks[L]
    with torch.no_grad():
        y_attn = blk.attn(x_mod)
        y_mlp = blk.mlp(x_mod)
    a_idx, a_aidx, m_idx = O.block_indices(12)[L]
    p = f"blocks.{L}."
    mine_attn = O.attn_as_written(x_mod, cp, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"],
                                  w[p + "attn.proj.weight"], w[p + "attn.proj.bias"], attn_idx=a_aidx,
                                  idx=a_idx, s=S, num_heads=12, scale=64 ** -0.5)
    mine_mlp = O.mlp_as_written(x_mod, cp, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"],
                                w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], idx=m_idx, s=S)
    assert torch.allclose(mine_attn, y_attn, rtol=1e-5, atol=1e-6), (mine_attn - y_attn).abs().max()
    assert torch.allclose(mine_mlp, y_mlp, rtol=1e-5, atol=1e-6), (mine_mlp - y_mlp).abs().max()
    out["mod_cfg"] = np.array([R, L, 101, 102, 103, 14], dtype=np.int64)
    out["mod_scale"] = np.array([S])
    out["mod_attn_out"] = y_attn.numpy().copy()
    out["mod_mlp_out"] = y_mlp.numpy().copy()

    # ---- 3. whole model, depth 12, 5 tokens (img 32): logits + CP grads, eval mode ----
    gx = torch.Generator(device="cpu").manual_seed(104)
    img = torch.randn(2, 3, 32, 32, generator=gx)
    m.zero_grad()
    logits = m(img)
    loss = torch.logsumexp(logits, dim=1).sum()
    loss.backward()
    out["full_cfg"] = np.array([R, 12, 32, 104], dtype=np.int64)
    out["full_logits"] = logits.detach().numpy().copy()
    for n in O.CP_NAMES:
        out["full_grad_" + n] = getattr(m, n).grad.detach().numpy().copy()
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    mine = O.vit_cara_forward(img, w, cpv, s=S, depth=12, num_heads=12)
    assert torch.allclose(mine, logits, rtol=1e-5, atol=1e-6), (mine - logits).abs().max()
    torch.logsumexp(mine, dim=1).sum().backward()
    for n in O.CP_NAMES:
        ga, gb = cpv[n].grad, getattr(m, n).grad
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-7), (n, (ga - gb).abs().max())
    # factored form (A.3) agrees with the as-written reference
    fac_logits = O.vit_cara_forward(img.double(), {k: v.double() for k, v in w.items()},
                                    {k: v.double() for k, v in cp.items()}, s=S, factored=True)
    assert torch.allclose(fac_logits.float(), logits.detach(), rtol=1e-4, atol=1e-5)

    # ---- 4. train mode: same RNG draw order (dropout on dW, DropPath) ----
    m.train()
    torch.manual_seed(77)
    with torch.no_grad():
        lt_ref = m(img)
    torch.manual_seed(77)
    dpr = [x.item() for x in torch.linspace(0, 0.1, 12)]
    with torch.no_grad():
        lt_mine = O.vit_cara_forward(img, w, cp, s=S, train={"dp": 0.1, "dpr": dpr})
    assert torch.allclose(lt_mine, lt_ref, rtol=1e-5, atol=1e-6), (lt_mine - lt_ref).abs().max()
    out["train_logits_seed77"] = lt_ref.numpy().copy()
    m.eval()

    # ---- 5. zero-init known-answer test: adapted == plain, bit for bit ----
    torch.manual_seed(5)
    plain = O.create_vit("vit_base_patch16_224_in21k", depth=2, num_classes=100, img_size=32)
    seeded_backbone_into(plain, 201)
    with torch.no_grad():
        base_logits = plain.eval()(img)
    adapted = ref.cara({"model": plain, "rank": 16, "scale": 1.0, "l_mu": 1.0, "l_std": 0.0}).eval()
    with torch.no_grad():
        kat_logits = adapted(img)
    assert torch.equal(base_logits, kat_logits)
    out["kat_cfg"] = np.array([2, 201, 104], dtype=np.int64)
    out["kat_logits"] = kat_logits.numpy().copy()

    # ---- 6. depth-2, 197-token model (real sequence length), rank 16, scale 0.1 ----
    torch.manual_seed(14)
    vit2 = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=2, num_classes=100)
    seeded_backbone_into(vit2, 301)
    m2 = ref.cara({"model": vit2, "rank": 16, "scale": 0.1, "l_mu": 1.5, "l_std": 0.1})
    randomise_cp(m2, 302)
    m2.eval()
    gx = torch.Generator(device="cpu").manual_seed(303)
    img2 = torch.randn(2, 3, 224, 224, generator=gx)
    lg2 = m2(img2)
    torch.logsumexp(lg2, dim=1).sum().backward()
    out["d2_cfg"] = np.array([16, 2, 224, 301, 302, 303, 14], dtype=np.int64)
    out["d2_logits"] = lg2.detach().numpy().copy()
    for n in O.CP_NAMES:
        out["d2_grad_" + n] = getattr(m2, n).grad.detach().numpy().copy()
    cp2 = cp_of(m2)
    mine2 = O.vit_cara_forward(img2, O.vit_weights(m2), cp2, s=0.1, depth=2)
    assert torch.allclose(mine2, lg2.detach(), rtol=1e-5, atol=1e-6)

    # ---- 7. the other orders of the QKV tensorisation: image_classification/dim_experiment.py (cp_length 3 and 5) ----
    # The script's own set_CP / cp_attn / cp_mlp (:186-346), loaded by file path; its module-level imports that are not
    # on this path (avalanche, wandb, torchvision, timm.scheduler) resolve to import-only stand-ins.  Depth 2, 197
    # tokens, rank 16, s = 0.1: index walk, init draws, logits and every CP gradient.
    dim = load_dim_experiment()
    for n_ in (3, 5):
        torch.manual_seed(14)
        vit3 = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=2, num_classes=100)
        seeded_backbone_into(vit3, 401)
        dim.vit = vit3          # the script keeps the model in a module global (dim_experiment.py:418)
        torch.manual_seed(15)
        dim.set_CP(vit3, dim=16, s=0.1, l_mu=1.5, l_std=0.1, cp_length=n_)
        names = [k for k, _ in vit3.named_parameters() if k.startswith("CP_")]
        torch.manual_seed(15)
        mine = O.init_cp_params(16, 1.5, 0.1, cp_length=n_)
        assert names == list(mine.keys()), (names, list(mine.keys()))
        for k in names:
            assert torch.equal(mine[k], getattr(vit3, k).detach()), (n_, k)      # same shapes, initialisers and draw order
        walk = [(b.attn.idx, b.attn.attn_idx, b.mlp.idx) for b in vit3.blocks]
        assert walk == [(0, 0, 1), (9, 1 if n_ == 5 else 3, 10)], walk
        g = torch.Generator(device="cpu").manual_seed(402)
        with torch.no_grad():       # non-zero adapters: the zero-initialised input factor of the QKV tensor and P2, the biases
            getattr(vit3, "CP_A3" if n_ == 5 else "CP_A2").copy_(0.05 * torch.randn(768, 16, generator=g))
            vit3.CP_P2.copy_(0.05 * torch.randn(768, 16, generator=g))
            for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
                getattr(vit3, k).copy_(0.02 * torch.randn(getattr(vit3, k).shape, generator=g))
        vit3.eval()
        img3 = torch.randn(2, 3, 224, 224, generator=torch.Generator(device="cpu").manual_seed(403))
        lg3 = vit3(img3)
        torch.logsumexp(lg3, dim=1).sum().backward()
        out[f"cpl{n_}_cfg"] = np.array([16, 2, 224, 401, 402, 403, 14, 15], dtype=np.int64)
        out[f"cpl{n_}_logits"] = lg3.detach().numpy().copy()
        cp3 = {k: getattr(vit3, k).detach().clone() for k in names}
        for k in names:
            out[f"cpl{n_}_grad_{k}"] = getattr(vit3, k).grad.detach().numpy().copy()
        cpv = {k: v.clone().requires_grad_(True) for k, v in cp3.items()}
        mine3 = O.vit_cara_forward(img3, O.vit_weights(vit3), cpv, s=0.1, depth=2)
        assert torch.allclose(mine3, lg3.detach(), rtol=1e-5, atol=5e-6), (n_, (mine3 - lg3).abs().max())   # fp32 summation order of the order-5 cp_to_tensor
        torch.logsumexp(mine3, dim=1).sum().backward()
        for k in names:
            assert torch.allclose(cpv[k].grad, getattr(vit3, k).grad, rtol=1e-4, atol=1e-7), (n_, k)

    # ---- 8. order 2 (dim_experiment.py:203-207,293-297): CP_A2 is [dim * dim, rank] -- `rank` dense dim x dim matrices per
    # projection.  Rank 4 keeps the vectors small; of the 589 824 x 4 gradient of CP_A2 every 97th row is recorded, with
    # the norm and the sum of the whole tensor.
    torch.manual_seed(14)
    vit2 = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=2, num_classes=100)
    seeded_backbone_into(vit2, 401)
    dim.vit = vit2
    torch.manual_seed(15)
    dim.set_CP(vit2, dim=4, s=0.1, l_mu=1.5, l_std=0.1, cp_length=2)
    names = [k for k, _ in vit2.named_parameters() if k.startswith("CP_")]
    torch.manual_seed(15)
    mine = O.init_cp_params(4, 1.5, 0.1, cp_length=2)
    assert names == list(mine.keys()), (names, list(mine.keys()))
    for k in names:
        assert torch.equal(mine[k], getattr(vit2, k).detach()), (2, k)
    assert [(b.attn.idx, b.attn.attn_idx, b.mlp.idx) for b in vit2.blocks] == [(0, 0, 1), (9, 3, 10)]
    g = torch.Generator(device="cpu").manual_seed(402)
    with torch.no_grad():
        vit2.CP_A2.copy_(0.02 * torch.randn(768 * 768, 4, generator=g))
        vit2.CP_P2.copy_(0.05 * torch.randn(768, 4, generator=g))
        for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
            getattr(vit2, k).copy_(0.02 * torch.randn(getattr(vit2, k).shape, generator=g))
    vit2.eval()
    img4 = torch.randn(2, 3, 224, 224, generator=torch.Generator(device="cpu").manual_seed(403))
    lg4 = vit2(img4)
    torch.logsumexp(lg4, dim=1).sum().backward()
    out["cpl2_cfg"] = np.array([4, 2, 224, 401, 402, 403, 14, 15], dtype=np.int64)
    out["cpl2_logits"] = lg4.detach().numpy().copy()
    for k in names:
        gk = getattr(vit2, k).grad.detach()
        if k == "CP_A2":
            out["cpl2_grad_CP_A2_rows97"] = gk[::97].numpy().copy()
            out["cpl2_grad_CP_A2_norm_sum"] = np.array([gk.double().norm().item(), gk.double().sum().item()])
        else:
            out[f"cpl2_grad_{k}"] = gk.numpy().copy()
    cp4 = {k: getattr(vit2, k).detach().clone() for k in names}
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp4.items()}
    mine4 = O.vit_cara_forward(img4, O.vit_weights(vit2), cpv, s=0.1, depth=2)
    assert torch.allclose(mine4, lg4.detach(), rtol=1e-5, atol=5e-6), (mine4 - lg4).abs().max()
    torch.logsumexp(mine4, dim=1).sum().backward()
    for k in names:
        assert torch.allclose(cpv[k].grad, getattr(vit2, k).grad, rtol=1e-4, atol=1e-7), (2, k)

    path = os.path.join(HERE, "cara_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
