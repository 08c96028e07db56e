"""CPU tests (no GPU): the oracle against the golden vectors recorded from the reference's own
``src/cara/cara.py`` (tests/golden/make_golden.py), plus the algebra the HIP path relies on
(SURVEY.md Appendix A.3 factored form, A.4 gradient identities)."""
import os

import numpy as np
import pytest
import torch

from oracle import cara_oracle as O
from tests.golden.inputs import oracle_case

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cara_reference_vectors.npz"))


def t(name):
    return torch.from_numpy(G[name])


def test_index_walk_matches_reference():
    # src/cara/cara.py:146-166 as executed by the reference
    assert G["idx_walk"].tolist() == [list(x) for x in O.block_indices(12)]
    assert G["final_idx"].tolist() == [108, 36]
    assert O.block_indices(12)[0] == (0, 0, 1) and O.block_indices(12)[1] == (9, 3, 10)


@pytest.mark.parametrize("rank", [8, 16, 32, 64])
def test_init_matches_reference(rank):
    # src/cara/cara.py:127-142 under torch.manual_seed(0), l_mu=1.5, l_std=0.1
    torch.manual_seed(0)
    cp = O.init_cp_params(rank, 1.5, 0.1)
    for n in ("CP_A1", "CP_A3", "CP_A4", "CP_P1", "CP_R1", "CP_R2"):
        assert torch.equal(cp[n], t(f"init_r{rank}_{n}")), n
    assert torch.equal(cp["CP_P3"][:8], t(f"init_r{rank}_CP_P3_rows0_8"))
    s = G[f"init_r{rank}_CP_P3_sum"]
    assert abs(cp["CP_P3"].double().sum().item() - s[0]) < 1e-9
    assert torch.count_nonzero(cp["CP_A2"]) == 0 and torch.count_nonzero(cp["CP_P2"]) == 0
    # shapes of cara.py:112-125
    assert cp["CP_A1"].shape == (36, rank) and cp["CP_P1"].shape == (108, rank)
    assert cp["CP_A4"].shape == (64, rank) and cp["CP_bias2"].shape == (3072,)


def test_lambda_ones():
    cp = O.init_cp_params(4, 1.0, 0.0)
    assert torch.equal(cp["CP_R1"], torch.ones(4)) and torch.equal(cp["CP_R2"], torch.ones(4))


@pytest.fixture(scope="module")
def case12():
    R, L, sb, sc, sx, sg = G["mod_cfg"].tolist()
    w, cp = oracle_case(sg, sb, sc, R, 12, 32)
    return R, L, float(G["mod_scale"][0]), w, cp, sx


def test_module_outputs_match_reference(case12):
    R, L, S, w, cp, sx = case12
    x = torch.randn(2, 7, 768, generator=torch.Generator().manual_seed(sx))
    a_idx, a_aidx, m_idx = O.block_indices(12)[L]
    p = f"blocks.{L}."
    ya = O.attn_as_written(x, cp, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"], w[p + "attn.proj.weight"],
                           w[p + "attn.proj.bias"], attn_idx=a_aidx, idx=a_idx, s=S, num_heads=12, scale=64 ** -0.5)
    ym = O.mlp_as_written(x, cp, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], w[p + "mlp.fc2.weight"],
                          w[p + "mlp.fc2.bias"], idx=m_idx, s=S)
    assert torch.allclose(ya, t("mod_attn_out"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(ym, t("mod_mlp_out"), rtol=1e-5, atol=1e-6)


def test_full_logits_and_grads_match_reference(case12):
    R, L, S, w, cp, _ = case12
    img = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(int(G["full_cfg"][3])))
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    logits = O.vit_cara_forward(img, w, cpv, s=S)
    assert torch.allclose(logits, t("full_logits"), rtol=1e-5, atol=1e-6)
    assert torch.equal(logits.argmax(1), t("full_logits").argmax(1))
    torch.logsumexp(logits, dim=1).sum().backward()
    for n in O.CP_NAMES:
        assert torch.allclose(cpv[n].grad, t("full_grad_" + n), rtol=1e-4, atol=1e-7), n


def test_factored_equals_as_written(case12):
    """A.3: delta = (x U) Vs^T + c_s reproduces the reference's materialised form."""
    R, L, S, w, cp, _ = case12
    img = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(int(G["full_cfg"][3])))
    d = lambda m: {k: v.double() for k, v in m.items()}  # noqa: E731
    fac = O.vit_cara_forward(img.double(), d(w), d(cp), s=S, factored=True)
    assert torch.allclose(fac.float(), t("full_logits"), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cp_length", [3, 5])
def test_other_orders_of_the_qkv_tensorisation_factor_the_same_way(cp_length):
    """dim_experiment.py's order-3 and order-5 QKV tensorisations are rank-R in (in, out) as well: the factored form the
    kernels run equals the materialised einsum of the script in fp64, logits and every CP gradient."""
    import torch
    w = {k: v.double() for k, v in O.synthetic_backbone(depth=2).items()}
    cp = {k: v.double() for k, v in O.synthetic_cp(rank=8, depth=2, cp_length=cp_length).items()}
    assert O.cp_length_of(cp) == cp_length
    x, y = O.synthetic_batch(batch=2)
    outs = []
    for factored in (False, True):
        cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
        logits = O.vit_cara_forward(x.double(), w, cpv, s=0.1, depth=2, factored=factored)
        torch.nn.functional.cross_entropy(logits, y).backward()
        outs.append((logits.detach(), {k: v.grad for k, v in cpv.items()}))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-10)
    for k in cp:
        assert torch.allclose(outs[0][1][k], outs[1][1][k], atol=1e-10), k
        assert k in ("CP_bias1", "CP_bias2", "CP_bias3") or outs[0][1][k].abs().max() > 0, k


def test_train_mode_rng_order(case12):
    """Dropout(0.1) on each materialised dW + DropPath, drawn in the reference's order."""
    R, L, S, w, cp, _ = case12
    img = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(int(G["full_cfg"][3])))
    dpr = [x.item() for x in torch.linspace(0, 0.1, 12)]
    torch.manual_seed(77)
    with torch.no_grad():
        lt = O.vit_cara_forward(img, w, cp, s=S, train={"dp": 0.1, "dpr": dpr})
    assert torch.allclose(lt, t("train_logits_seed77"), rtol=1e-5, atol=1e-6)


def test_zero_init_known_answer():
    """CP_A2 = CP_P2 = 0 (tests/test_cara.py:79-83) => adapted logits == plain logits exactly."""
    depth, sb, sx = G["kat_cfg"].tolist()
    torch.manual_seed(5)
    from tests.golden.inputs import seeded_backbone_into
    plain = O.create_vit("vit_base_patch16_224_in21k", depth=depth, num_classes=100, img_size=32)
    seeded_backbone_into(plain, sb)
    img = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(sx))
    with torch.no_grad():
        base = plain.eval()(img)
    assert torch.equal(base, t("kat_logits"))
    cp = O.init_cp_params(16, 1.0, 0.0)
    with torch.no_grad():
        mine = O.vit_cara_forward(img, O.vit_weights(plain), cp, s=1.0, depth=depth)
    assert torch.allclose(mine, base, rtol=0, atol=2e-6)  # functional path differs only by conv->GEMM order
    assert torch.equal(mine.argmax(1), base.argmax(1))


def test_depth2_197_tokens_match_reference():
    R, depth, imgsz, sb, sc, sx, sg = G["d2_cfg"].tolist()
    w, cp = oracle_case(sg, sb, sc, R, depth, imgsz)
    img = torch.randn(2, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(sx))
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    logits = O.vit_cara_forward(img, w, cpv, s=0.1, depth=depth)
    assert torch.allclose(logits, t("d2_logits"), rtol=1e-5, atol=1e-6)
    torch.logsumexp(logits, dim=1).sum().backward()
    for n in O.CP_NAMES:
        assert torch.allclose(cpv[n].grad, t("d2_grad_" + n), rtol=1e-4, atol=1e-7), n


def test_cp_to_tensor_definition():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(5, generator=g, dtype=torch.float64)
    fs = [torch.randn(n, 5, generator=g, dtype=torch.float64) for n in (3, 4, 2, 6)]
    ref = torch.einsum("r,ar,br,cr,dr->abcd", w, *fs)
    assert torch.allclose(O.cp_to_tensor((w, fs)), ref, atol=1e-12)
    assert O.cp_to_tensor((w, fs)).shape == (3, 4, 2, 6)


def test_a4_gradient_identities():
    """A.4: every factor gradient follows from two skinny products per linear,
    dU = X^T G' and dVs = dY^T T (G' = dY Vs, T = X U), plus dc = sum dY."""
    torch.manual_seed(0)
    R, M, s = 8, 10, 0.1
    cp = {k: torch.randn(*sh, dtype=torch.float64) * 0.3 for k, sh in O.cp_shapes(R).items()}
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    fac = O.build_factored(cpv, s)[5]
    for name, din, dout in (("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)):
        U, Vs, cs = fac[name]
        x = torch.randn(M, din, dtype=torch.float64)
        dy = torch.randn(M, dout, dtype=torch.float64)
        y = (x @ U) @ Vs.t() + (cs if cs is not None else 0)
        gU, gV = torch.autograd.grad(y, (U, Vs), dy, retain_graph=True)
        assert torch.allclose(gU, x.t() @ (dy @ Vs.detach()), atol=1e-10)
        assert torch.allclose(gV, dy.t() @ (x @ U.detach()), atol=1e-10)
        if cs is not None:
            (gc,) = torch.autograd.grad(y, (cs,), dy, retain_graph=True)
            assert torch.allclose(gc, dy.sum(0), atol=1e-10)


@pytest.mark.parametrize("cp_length", [3, 5])
def test_other_orders_match_the_reference_script(cp_length):
    """Golden case 7: logits and every CP gradient recorded from the reference's OWN image_classification/
    dim_experiment.py (set_CP / cp_attn / cp_mlp with cp_length 3 and 5) -- the oracle's restatement of those
    tensorisations, as-written and factored, against them."""
    from tests.golden.inputs import oracle_case_cp_length
    w, cp, img = oracle_case_cp_length(cp_length)
    ref = torch.from_numpy(G[f"cpl{cp_length}_logits"])
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    logits = O.vit_cara_forward(img, w, cpv, s=0.1, depth=2)
    assert torch.allclose(logits, ref, rtol=1e-5, atol=5e-6)
    torch.logsumexp(logits, dim=1).sum().backward()
    for k in cp:
        assert torch.allclose(cpv[k].grad, torch.from_numpy(G[f"cpl{cp_length}_grad_{k}"]), rtol=1e-4, atol=1e-7), k
    fac = O.vit_cara_forward(img.double(), {k: v.double() for k, v in w.items()}, {k: v.double() for k, v in cp.items()}, s=0.1,
                             depth=2, factored=True)
    assert torch.allclose(fac.float(), ref, rtol=1e-4, atol=1e-5)


def test_order_2_matches_the_reference_script():
    """Golden case 8: the order-2 tensorisation (dim_experiment.py:203-207,293-297: CP_A2 [dim * dim, rank], a sum of `rank`
    dense matrices per projection) -- logits, every small CP gradient, and of the 589 824 x 4 gradient of CP_A2 every 97th
    row with the norm and sum of the whole, all recorded from the reference's own script; the oracle's "dense delta" form (two
    products on the same operand: what the device runs) equals the as-written einsum."""
    from tests.golden.inputs import oracle_case_cp_length2
    w, cp, img = oracle_case_cp_length2()
    assert O.cp_length_of(cp) == 2 and cp["CP_A2"].shape == (768 * 768, 4)
    ref = torch.from_numpy(G["cpl2_logits"])
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    logits = O.vit_cara_forward(img, w, cpv, s=0.1, depth=2)
    assert torch.allclose(logits, ref, rtol=1e-5, atol=5e-6)
    torch.logsumexp(logits, dim=1).sum().backward()
    for k in cp:
        if k == "CP_A2":
            g2 = cpv[k].grad
            assert torch.allclose(g2[::97], torch.from_numpy(G["cpl2_grad_CP_A2_rows97"]), rtol=1e-4, atol=1e-8)
            n, sm = G["cpl2_grad_CP_A2_norm_sum"]
            assert abs(g2.double().norm().item() - n) < 1e-5 * n and abs(g2.double().sum().item() - sm) < 1e-4 * abs(n)
        else:
            assert torch.allclose(cpv[k].grad, torch.from_numpy(G[f"cpl2_grad_{k}"]), rtol=1e-4, atol=1e-7), k
    fac = O.vit_cara_forward(img.double(), {k: v.double() for k, v in w.items()}, {k: v.double() for k, v in cp.items()}, s=0.1,
                             depth=2, factored=True)
    assert torch.allclose(fac.float(), ref, rtol=1e-4, atol=1e-5)
