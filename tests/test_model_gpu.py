"""GPU parity of the whole adapted ViT (cara_vit_forward / cara_vit_backward through the Python
mirror of the reference API) against (a) vectors recorded from the reference's own cara.py and
(b) the CPU oracle on the same seeded inputs.

Tolerances: tests/tolerances.py states the contract once.  precision = "bf16" (the build BASELINE.json's metric is quoted on):
logits <= 1.15 x the error of the oracle evaluated with the same bf16 rounding points and <= 1e-2, every CP gradient <= 2.5e-2,
one block vs the bf16-rounded oracle <= 1e-3, class indices exact where the fp32 margin exceeds the logit noise -- north_star's
1e-3 on the logits is below the rounding floor of 8-bit-significand MFMA operands and is NOT met by that
build.  precision = "fp16" (the same kernels with IEEE-half operands, same MFMA rate): north_star's numbers as written -- logits
<= 1e-3 rel-L2 of the fp32 reference, EVERY class index equal, CP gradients <= 5e-3 -- asserted on every configuration the
whole-model tests run (`PRECISIONS` below).  Measured values are printed (-s) and recorded in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch

from tests import tolerances as T

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cara_reference_vectors.npz"))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


PRECISIONS = ["bf16", "fp16"]
FP16_LOGITS = 1.0e-3   # north_star: "within 1e-3 relative on ... logits", against the fp32 reference
FP16_CP_GRAD = 0.2 * T.CP_GRAD


def check_logits(logits, ref, sim, precision, what=""):
    """The parity bar of a whole-model case, per precision (module docstring): returns the measured rel-L2."""
    r_ref = rel(logits, ref)
    if precision == "fp16":
        differ = int((logits.argmax(1).cpu() != ref.argmax(1)).sum())
        print(f"{what} [fp16]: logits rel-L2 vs the fp32 reference {r_ref:.2e}; class indices that differ: {differ} of {ref.shape[0]}")
        assert r_ref <= FP16_LOGITS, (what, r_ref)
        assert differ == 0, (what, differ)
    else:
        r_model = rel(sim, ref)
        print(f"{what} [bf16]: logits rel-L2 vs the fp32 reference {r_ref:.2e} (rounding model {r_model:.2e})")
        assert T.logits_ok(r_ref, r_model), (what, r_ref, r_model)
        top2 = ref.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 4 * (logits.detach().cpu() - ref).abs().max()
        assert torch.equal(logits.argmax(1).cpu()[safe], ref.argmax(1)[safe]), what
    return r_ref


def grad_bar(precision):
    return FP16_CP_GRAD if precision == "fp16" else T.CP_GRAD


def build(w, cp, rank, scale, depth, img, num_classes=100, drop_path_rate=0.1, name="vit_base_patch16_224_in21k", cp_length=4,
          precision="bf16"):
    from cara_amd import cara, create_model
    m = create_model(name, drop_path_rate=drop_path_rate, depth=depth, img_size=img, num_classes=num_classes)
    m = cara({"model": m, "rank": rank, "scale": scale, "l_mu": 1.5, "l_std": 0.1, "cp_length": cp_length, "precision": precision})
    sd = dict(w)
    sd.update(cp)
    # the reference hard-codes 36 / 108 rows (cara.py:112,118 = 3 / 9 per block at depth 12); this
    # build sizes them 3*depth / 9*depth, and a shallower test model only ever reads its own rows
    # (order 5: one CP_A1 row per block, dim_experiment.py:266)
    sd["CP_A1"], sd["CP_P1"] = cp["CP_A1"][:(1 if cp_length == 5 else 3) * depth], cp["CP_P1"][:9 * depth]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.to(DEV)


def test_reference_structural_forward_shape():
    """/root/reference/tests/test_cara.py:93-98: [2,3,224,224] -> (2, 21843), train-mode default."""
    from cara_amd import cara, create_model
    torch.manual_seed(0)
    vit = cara({"model": create_model("vit_base_patch16_224_in21k", drop_path_rate=0.1), "rank": 32, "scale": 1.0,
                "l_mu": 1.0, "l_std": 0.0}).to(DEV)
    out = vit(torch.randn(2, 3, 224, 224, device=DEV))
    assert tuple(out.shape) == (2, 21843) and torch.isfinite(out).all()


def test_depth2_against_reference_vectors():
    """Golden case 6 of make_golden.py: depth 2, 197 tokens, rank 16, s = 0.1 -- logits and all 12 CP
    gradients of sum(logsumexp(logits)) as produced by the reference's cara.py."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case
    R, depth, imgsz, sb, sc, sx, sg = G["d2_cfg"].tolist()
    w, cp = oracle_case(sg, sb, sc, R, depth, imgsz)
    m = build(w, cp, R, 0.1, depth, imgsz).eval()
    img = torch.randn(2, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(sx))
    logits = m(img.to(DEV))
    ref = torch.from_numpy(G["d2_logits"])
    with torch.no_grad():
        sim = O.vit_cara_forward(img, w, cp, s=0.1, depth=depth, factored=True, bf16_sim=True)
    r_ref, r_sim = rel(logits, ref), rel(logits, sim)
    print(f"depth2 logits rel-L2: vs fp32 reference {r_ref:.2e}, vs bf16-rounded oracle {r_sim:.2e}")
    r_model = rel(sim, ref)   # what bf16 rounding at the same points costs, per the oracle
    assert T.logits_ok(r_ref, r_model), (r_ref, r_model)
    assert r_sim < T.LOGITS_VS_MODEL * max(r_model, T.MODEL_FLOOR), (r_sim, r_model)
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    torch.logsumexp(logits, dim=1).sum().backward()
    worst = 0.0
    for n in O.CP_NAMES:
        g, gr = getattr(m, n).grad, torch.from_numpy(G["d2_grad_" + n])
        if n in ("CP_A1", "CP_P1"):
            assert torch.count_nonzero(gr[g.shape[0]:]) == 0   # rows of blocks that do not exist
            gr = gr[:g.shape[0]]
        r = rel(g, gr)
        worst = max(worst, r)
        assert r < T.CP_GRAD, (n, r)     # bf16 activations/gradients vs fp32 autograd of the dense form
    print(f"depth2 CP-gradient worst rel-L2 vs reference: {worst:.2e}")


def test_exact_weight_dropout_mode_against_oracle():
    """engine.weight_dropout = "exact": the reference's TRAIN-mode arithmetic (Dropout(0.1) on every materialised
    dW, cara.py:35,57,81,92) with the masks rebuilt on the CPU from the same counter hash and fed to the oracle's
    as-written algorithm: logits, loss and all 12 CP gradients.  depth 3, batch 4 (788 rows: the zero-padded K of
    the dense dW GEMM), s = 1 so that the dropped elements matter; DropPath off to isolate the weight dropout."""
    from oracle import cara_oracle as O
    from cara_amd.dropout import keep_mask
    torch.manual_seed(0)
    depth, R, s_, p, seed = 3, 16, 1.0, 0.1, 4242
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=R)
    x, y = O.synthetic_batch(batch=4)
    m = build(w, cp, R, s_, depth, 224, drop_path_rate=0.0).train()
    eng = m._cara_engine
    eng.weight_dropout, eng.weight_dropout_p, eng.weight_dropout_seed = "exact", p, seed
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, y.to(DEV))
    loss.backward()
    dims = {"qkv": (2304, 768, 0), "proj": (768, 768, 1), "fc1": (3072, 768, 2), "fc2": (768, 3072, 3)}

    def masks(layer, name):
        o, i, slot = dims[name]
        return torch.from_numpy(keep_mask(o, i, p, seed, 4 * layer + slot))

    cps = dict(cp)
    cps["CP_A1"], cps["CP_P1"] = cp["CP_A1"][:3 * depth], cp["CP_P1"][:9 * depth]
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    rloss, rlogits, gref = O.train_step_as_written(x, y, w, cps, head, s=s_, depth=depth, keep_masks=masks, keep_p=p)
    r = rel(logits, rlogits)
    # what the masks are worth: the same forward with no dropout, and with another seed
    with torch.no_grad():
        nodrop = O.vit_cara_forward(x, w, cps, s=s_, depth=depth)
    eng.weight_dropout_seed = seed + 1
    with torch.no_grad():
        other = m(x.to(DEV))
    print(f"exact weight dropout: logits rel-L2 vs oracle with the same masks {r:.2e}; masks move the logits by "
          f"{rel(rlogits, nodrop):.2e}; another seed by {rel(other, rlogits):.2e}; loss {loss.item():.4f} vs {rloss.item():.4f}")
    assert r < T.LOGITS_ABS and rel(rlogits, nodrop) > 5 * r and rel(other, rlogits) > 5 * r
    assert abs(loss.item() - rloss.item()) < 2e-2 * max(1.0, abs(rloss.item()))
    worst = 0.0
    for n in O.CP_NAMES:
        rr = rel(getattr(m, n).grad, gref[n])
        worst = max(worst, rr)
        assert rr < T.CP_GRAD, (n, rr)
    print(f"exact weight dropout: worst CP-gradient rel-L2 vs fp32 autograd with the same masks {worst:.2e}")
    # eval is the factored path whatever the mode: bitwise the same logits as an engine with weight_dropout off
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV))
    eng.weight_dropout = "off"
    with torch.no_grad():
        ev_off = m(x.to(DEV))
    assert torch.equal(ev, ev_off)


def test_vit_large_384_against_oracle():
    """BASELINE.json configs[4] dimensioning: ViT-L/16 @384 -- dim 1024, 16 heads, 24 blocks, 577 tokens (the
    two-sweep attention path), CP_A1 [72,R], CP_A3 [16,R], CP_P1 [216,R], CP_A2/P2/P3 [1024,R], biases
    1024/4096/1024.  Batch 2, rank 16: logits and every CP gradient against the fp32 oracle
    (as-written dense-dW algorithm) and the rounding model."""
    from oracle import cara_oracle as O
    torch.manual_seed(0)
    dims = dict(depth=24, dim=1024, heads=16)
    w = O.synthetic_backbone(img=384, **dims)
    cp = O.synthetic_cp(rank=16, **dims)
    x, y = O.synthetic_batch(batch=2, img=384)
    m = build(w, cp, 16, 0.1, 24, 384, name="vit_large_patch16_384").eval()
    assert m.CP_A1.shape == (72, 16) and m.CP_A3.shape == (16, 16) and m.CP_P1.shape == (216, 16)
    assert m.CP_P2.shape == (1024, 16) and m.CP_bias2.shape == (4096,) and m.idx == 216 and m.attn_idx == 72
    logits = m(x.to(DEV))
    # expected values from the committed fixture (the oracle on these very inputs, tests/golden/make_headline_fixtures.py vitl);
    # recomputed on the spot only when the fixture is missing
    fx = os.path.join(os.path.dirname(__file__), "golden", "vit_large_384_b2_r16.npz")
    F_ = np.load(fx) if os.path.exists(fx) else None
    if F_ is not None:
        ref, sim = torch.from_numpy(F_["logits"]), torch.from_numpy(F_["logits_bf16_sim"])
    else:
        with torch.no_grad():
            ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16)
            sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16, factored=True, bf16_sim=True)
    r_ref, r_sim, r_model = rel(logits, ref), rel(logits, sim), rel(sim, ref)
    print(f"ViT-L/16@384 logits rel-L2: vs fp32 oracle {r_ref:.2e}, vs bf16-rounded oracle {r_sim:.2e} (rounding model {r_model:.2e})")
    assert T.logits_ok(r_ref, r_model), (r_ref, r_model)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * (logits.cpu() - ref).abs().max()
    assert torch.equal(logits.argmax(1).cpu()[safe], ref.argmax(1)[safe])
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    if F_ is not None:
        gref = {k[len("grad_"):]: torch.from_numpy(F_[k]) for k in F_.files if k.startswith("grad_")}
    else:
        head = {"weight": w["head.weight"], "bias": w["head.bias"]}
        _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1, depth=24, num_heads=16)
    worst = 0.0
    for n in O.CP_NAMES:
        r = rel(getattr(m, n).grad, gref[n])
        worst = max(worst, r)
        assert r < T.CP_GRAD, (n, r)
    print(f"ViT-L/16@384 worst CP-gradient rel-L2 vs fp32 autograd of the as-written form: {worst:.2e}")
    assert rel(m.head.weight.grad, gref["head.weight"]) < T.CP_GRAD


def test_vit_large_384_at_its_real_batch():
    """BASELINE.json configs[4] at its REAL per-GPU batch: ViT-L/16 @384, 32 images -> M = 18 464 token rows (115.4 row tiles of
    160, 144.25 of 128: the 160 x 256 x 64 tile's edge workgroups, the two-sweep attention at 32 x 16 heads), rank 16, eval mode.
    Logits against the fp32 as-written oracle and its bf16-rounded form, committed by tests/golden/make_headline_fixtures.py vitl32
    (115 s of CPU); the class index of every sample, no margin filter; and the same with precision = "fp16"."""
    from oracle import cara_oracle as O
    fx = os.path.join(os.path.dirname(__file__), "golden", "vit_large_384_b32_r16_logits.npz")
    if not os.path.exists(fx):
        pytest.skip("the committed ViT-L batch-32 fixture is missing")
    F_ = np.load(fx)
    ref, sim = torch.from_numpy(F_["logits"]), torch.from_numpy(F_["logits_bf16_sim"])
    dims = dict(depth=24, dim=1024, heads=16)
    w = O.synthetic_backbone(img=384, **dims)
    cp = O.synthetic_cp(rank=16, **dims)
    x, _ = O.synthetic_batch(batch=32, img=384)
    m = build(w, cp, 16, 0.1, 24, 384, name="vit_large_patch16_384").eval()
    with torch.no_grad():
        logits = m(x.to(DEV))
        m._cara_engine.precision = "fp16"
        half = m(x.to(DEV))
    r_ref, r_model, r_half = rel(logits, ref), rel(sim, ref), rel(half, ref)
    differ = int((logits.argmax(1).cpu() != ref.argmax(1)).sum()), int((half.argmax(1).cpu() != ref.argmax(1)).sum())
    print(f"\nViT-L/16@384, batch 32 (M = 18464): logits rel-L2 vs fp32 oracle bf16 {r_ref:.2e} (rounding model {r_model:.2e}), fp16 {r_half:.2e}; "
          f"class indices that differ of 32: bf16 {differ[0]}, fp16 {differ[1]}")
    assert T.logits_ok(r_ref, r_model), (r_ref, r_model)
    assert r_half < 0.25 * r_ref and r_half <= FP16_LOGITS   # north_star's number, 24 blocks deep (measured 9.0e-4 in r04, before the fp32 head)
    assert differ[1] == 0


def test_vit_large_384_train_step_at_its_real_batch():
    """BASELINE.json configs[4]'s TRAIN step at its real per-GPU batch (ViT-L/16 @384, 32 images, M = 18 464 rows: edge tiles of every
    GEMM family, the long-sequence attention forward AND backward over 32 x 16 heads, riders at 18 464 rows).  fp32 autograd of the
    as-written algorithm at this size does not fit the build container (the oracle checks logits at b32 and every gradient at b2:
    the two tests above), so the train step is pinned by a size-independent property instead: the mean cross-entropy over 32 images is
    the mean of the means over its sixteen pairs, so train_step(b32) must give the average of sixteen train_step(b2) -- the
    oracle-checked size -- loss and every gradient, up to fp32 summation order of the row reductions (no DropPath: eval mode)."""
    from oracle import cara_oracle as O
    dims = dict(depth=24, dim=1024, heads=16)
    w = O.synthetic_backbone(img=384, **dims)
    cp = O.synthetic_cp(rank=16, **dims)
    x, y = O.synthetic_batch(batch=32, img=384)
    m = build(w, cp, 16, 0.1, 24, 384, name="vit_large_patch16_384").eval()
    eng = m._cara_engine
    xd, yd = x.to(DEV), y.to(DEV)
    big = eng.train_step(xd, yd, None).item()
    g_big = eng._flat_grad[:-1].clone()
    acc, losses = torch.zeros_like(g_big), []
    for i in range(16):
        losses.append(eng.train_step(xd[2 * i:2 * i + 2], yd[2 * i:2 * i + 2], None).item())
        acc += eng._flat_grad[:-1]
    acc /= 16
    r = rel(g_big, acc)
    print(f"\nViT-L/16@384 train step, batch 32 vs the mean of sixteen batch-2 steps: loss {big:.6f} vs {sum(losses) / 16:.6f}; flat gradient rel-L2 {r:.2e}")
    # (per-row arithmetic does not depend on the batch: what may differ is the kernel family a product lands on -- the 160-row tile
    # at 18 464 rows, the 128-tile family at 1 154 -- and the order of the fp32 row reductions; 2e-3 is five times tighter than the
    # bar the b2 gradients are held to against fp32 autograd)
    assert abs(big - sum(losses) / 16) < 1e-4 * max(1.0, abs(big))
    assert r < 2e-3, r
    assert torch.isfinite(g_big).all() and g_big.abs().max() > 0


def test_depth12_headline_shapes_against_oracle():
    """ViT-B/16 depth 12, rank 16, 197 tokens, synthetic weights of SURVEY 8(d), batch 4."""
    from oracle import cara_oracle as O
    torch.manual_seed(0)
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=16)
    x, y = O.synthetic_batch(batch=4)
    m = build(w, cp, 16, 0.1, 12, 224).eval()
    logits = m(x.to(DEV))
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, factored=True, bf16_sim=True)
    r_ref, r_sim = rel(logits, ref), rel(logits, sim)
    print(f"depth12 logits rel-L2: vs fp32 oracle {r_ref:.2e}, vs bf16-rounded oracle {r_sim:.2e}")
    r_model = rel(sim, ref)
    assert T.logits_ok(r_ref, r_model), (r_ref, r_model)
    assert r_sim < T.LOGITS_VS_MODEL * max(r_model, T.MODEL_FLOOR), (r_sim, r_model)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * (logits.cpu() - ref).abs().max()
    assert torch.equal(logits.argmax(1).cpu()[safe], ref.argmax(1)[safe])
    # gradients of the training loss against fp32 autograd of the as-written algorithm
    loss = torch.nn.functional.cross_entropy(logits, y.to(DEV))
    loss.backward()
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1)
    for n in O.CP_NAMES:
        r = rel(getattr(m, n).grad, gref[n][:getattr(m, n).shape[0]])
        assert r < T.CP_GRAD, (n, r)
    assert rel(m.head.weight.grad, gref["head.weight"]) < T.CP_GRAD
    assert rel(m.head.bias.grad, gref["head.bias"]) < T.CP_GRAD


def test_zero_init_known_answer_bitwise():
    """CP_A2 = CP_P2 = 0 (reference tests/test_cara.py:79-83) => the adapter contributes exactly
    nothing: logits are bitwise independent of every other CP tensor."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone(depth=2)
    x, _ = O.synthetic_batch(batch=2)
    outs = []
    for seed, rank in ((14, 16), (99, 8)):
        cp = O.synthetic_cp(rank=rank, seed=seed)
        cp["CP_A2"].zero_(); cp["CP_P2"].zero_()
        for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
            cp[k].zero_()
        outs.append(build(w, cp, rank, 1.0, 2, 224).eval()(x.to(DEV)).detach())
    assert torch.equal(outs[0], outs[1])
    with torch.no_grad():
        base = O.vit_cara_forward(x, w, O.init_cp_params(4, 1.0, 0.0), s=1.0, depth=2)
    assert rel(outs[0], base) < 1e-2 and torch.equal(outs[0].argmax(1).cpu(), base.argmax(1))


def test_inference_forward_keeps_nothing_for_backward():
    """Under no_grad the forward runs with cara_vit_shape.inference = 1 (no pre-activation kept): the logits are
    bitwise those of the training-capable forward, and a backward cannot be started from it."""
    from oracle import cara_oracle as O
    from cara_amd._lib import CaraError
    w = O.synthetic_backbone(depth=3)
    cp = O.synthetic_cp(rank=16)
    x, _ = O.synthetic_batch(batch=8)       # 8 x 197 rows: the full-size GEMM path, not the few-row one
    m = build(w, cp, 16, 0.1, 3, 224).eval()
    eng = m._cara_engine
    a = m(x.to(DEV))                         # grad mode on: everything kept
    assert eng._bwd_ready == eng._fwd_serial
    with torch.no_grad():
        b = m(x.to(DEV))
    assert eng._bwd_ready == -1 and torch.equal(a.detach(), b)
    with pytest.raises(CaraError):
        a.sum().backward()                   # the workspace now holds the inference forward


@pytest.mark.parametrize("cp_length", [3, 5])
def test_other_orders_of_the_qkv_tensorisation_against_oracle(cp_length):
    """cp_length 3 and 5 of image_classification/dim_experiment.py (QKV adapter as an order-3 / order-5 CP tensor):
    same kernels, another factor pack and gradient scatter.  Logits against the fp32 as-written oracle within the
    bf16 rounding model, every CP gradient (A1..A3 / A1..A5 included) against autograd of the as-written form."""
    from oracle import cara_oracle as O
    depth, rank, B = 3, 16, 8
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=rank, cp_length=cp_length)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, rank, 0.1, depth, 224, drop_path_rate=0.0, cp_length=cp_length).train()
    assert m._cara_engine.cp_length == cp_length and ("CP_A5" in dict(m.named_parameters())) == (cp_length == 5)
    assert ("CP_A4" in dict(m.named_parameters())) == (cp_length != 3)
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, y.to(DEV))
    loss.backward()
    cpo = {k: (v[:(1 if cp_length == 5 else 3) * depth] if k == "CP_A1" else (v[:9 * depth] if k == "CP_P1" else v)) for k, v in cp.items()}
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    rloss, rlogits, rg = O.train_step_as_written(x, y, w, cpo, head, s=0.1, depth=depth)
    with torch.no_grad():
        sim = O.vit_cara_forward(x, w, cpo, s=0.1, depth=depth, factored=True, bf16_sim=True)
    e = rel(logits.detach(), rlogits)
    print(f"cp_length {cp_length}: logits rel {e:.2e} (bf16 model {rel(sim, rlogits):.2e})")
    assert T.logits_ok(e, rel(sim, rlogits)), (e, rel(sim, rlogits))
    for k in cpo:
        gk = getattr(m, k).grad
        assert gk is not None and torch.isfinite(gk).all(), k
        if rg[k].norm() > 0:
            ek = rel(gk, rg[k])
            print(f"  d{k}: rel {ek:.2e}")
            assert ek < T.CP_GRAD, (k, ek)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cp_length", [3, 5])
def test_other_orders_against_the_reference_script_vectors(cp_length, precision):
    """Golden case 7 of make_golden.py: logits and CP gradients recorded from the reference's own
    image_classification/dim_experiment.py (cp_length 3 and 5; depth 2, 197 tokens, rank 16) -- the device path
    against what that script computed."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case_cp_length
    w, cp, img = oracle_case_cp_length(cp_length)
    m = build(w, {k: v.clone() for k, v in cp.items()}, 16, 0.1, 2, 224, cp_length=cp_length, precision=precision).eval()
    logits = m(img.to(DEV))
    ref = torch.from_numpy(G[f"cpl{cp_length}_logits"])
    with torch.no_grad():
        sim = O.vit_cara_forward(img, w, cp, s=0.1, depth=2, factored=True, bf16_sim=True)
    check_logits(logits, ref, sim, precision, f"cp_length {cp_length} vs dim_experiment.py")
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    torch.logsumexp(logits, dim=1).sum().backward()
    worst = 0.0
    for k in cp:
        g, gr = getattr(m, k).grad, torch.from_numpy(G[f"cpl{cp_length}_grad_{k}"])
        if k in ("CP_A1", "CP_P1"):
            assert torch.count_nonzero(gr[g.shape[0]:]) == 0     # rows of blocks that do not exist at depth 2
            gr = gr[:g.shape[0]]
        worst = max(worst, rel(g, gr))
    print(f"cp_length {cp_length} [{precision}]: worst CP-gradient rel-L2 vs the script {worst:.2e}")
    assert worst < grad_bar(precision)


def test_order_2_against_the_reference_script_vectors():
    """Golden case 8 of make_golden.py: order 2 of image_classification/dim_experiment.py (`--dims 2`: every QKV projection
    gets a sum of R DENSE dim x dim matrices, CP_A2 [dim * dim, R], :203-207, :293-297) at depth 2, 197 tokens, rank 4 --
    the dense-delta device path against the logits and CP gradients that script computed.  CP_A2's gradient is 590k x 4:
    the fixture holds every 97th row plus the tensor's norm and sum."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case_cp_length2
    from cara_amd import CaraError
    w, cp, img = oracle_case_cp_length2()
    m = build(w, {k: v.clone() for k, v in cp.items()}, 4, 0.1, 2, 224, cp_length=2).eval()
    assert tuple(m.CP_A2.shape) == (768 * 768, 4) and not hasattr(m, "CP_A3")
    logits = m(img.to(DEV))
    ref = torch.from_numpy(G["cpl2_logits"])
    with torch.no_grad():
        sim = O.vit_cara_forward(img, w, cp, s=0.1, depth=2, factored=True, bf16_sim=True)
    r_ref, r_model = rel(logits, ref), rel(sim, ref)
    print(f"cp_length 2 vs dim_experiment.py: logits rel-L2 {r_ref:.2e} (rounding model {r_model:.2e})")
    assert T.logits_ok(r_ref, r_model), (r_ref, r_model)
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    torch.logsumexp(logits, dim=1).sum().backward()
    worst = 0.0
    for k in cp:
        g = getattr(m, k).grad
        if k == "CP_A2":
            e = rel(g[::97], torch.from_numpy(G["cpl2_grad_CP_A2_rows97"]))
            nrm, tot = G["cpl2_grad_CP_A2_norm_sum"].tolist()
            assert abs(g.double().norm().item() / nrm - 1) < T.CP_GRAD
            assert abs(g.double().sum().item() - tot) < T.CP_GRAD * nrm          # a sum of 2.4 M terms of norm `nrm`
        else:
            gr = torch.from_numpy(G[f"cpl2_grad_{k}"])
            if k in ("CP_A1", "CP_P1"):
                assert torch.count_nonzero(gr[g.shape[0]:]) == 0
                gr = gr[:g.shape[0]]
            e = rel(g, gr)
        print(f"  cp_length 2 grad {k}: rel-L2 {e:.2e}")
        worst = max(worst, e)
    assert worst < T.CP_GRAD
    # the sub-module entries carry the factored adapters only
    with pytest.raises(CaraError, match="whole model"):
        m.blocks[0].attn(torch.zeros(1, 197, 768, device=DEV))


def test_order_2_train_step_and_deeper_model():
    """Order 2 through the optimiser step, in train mode (drop-path masks), at depth 3 / rank 8 / odd batch, against the oracle's
    autograd on the same masks; then two AdamW steps lower the loss."""
    from oracle import cara_oracle as O
    depth, rank = 3, 8
    w = O.synthetic_backbone(depth=depth)
    torch.manual_seed(5)
    cp = O.init_cp_params(rank, 1.0, 0.1, depth=depth, cp_length=2)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        cp["CP_A2"].copy_(0.02 * torch.randn(cp["CP_A2"].shape, generator=g))
        cp["CP_P2"].copy_(0.05 * torch.randn(cp["CP_P2"].shape, generator=g))
    x, y = O.synthetic_batch(batch=3)
    m = build(w, {k: v.clone() for k, v in cp.items()}, rank, 0.1, depth, 224, cp_length=2).train()
    keep = torch.ones(depth, 2, 3)
    keep[1, 0, 1] = 0.0
    keep[2, 1, 2] = 0.0
    logits = m._cara_engine.forward(x.to(DEV), droppath=keep.to(DEV))
    cpv = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    ref = O.vit_cara_forward(x, w, cpv, s=0.1, depth=depth, drop_path_keep=keep)
    with torch.no_grad():
        sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=depth, drop_path_keep=keep, factored=True, bf16_sim=True)
    assert T.logits_ok(rel(logits, ref.detach()), rel(sim, ref.detach()))
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    torch.nn.functional.cross_entropy(ref, y).backward()
    for k in cp:
        e = rel(getattr(m, k).grad, cpv[k].grad)
        print(f"  order 2, depth 3: grad {k} rel-L2 {e:.2e}")
        assert e < T.CP_GRAD, (k, e)
    m.zero_grad()
    m.eval()
    opt = torch.optim.AdamW([p for n, p in m.named_parameters() if n.startswith("CP_") or n.startswith("head")], lr=1e-3)
    losses = []
    for _ in range(3):
        loss = torch.nn.functional.cross_entropy(m(x.to(DEV)), y.to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def test_drop_path_masks_and_train_mode():
    from oracle import cara_oracle as O
    w = O.synthetic_backbone(depth=3)
    cp = O.synthetic_cp(rank=16)
    x, _ = O.synthetic_batch(batch=4)
    m = build(w, cp, 16, 0.1, 3, 224).train()
    keep = torch.tensor([[[1, 1, 1, 1], [1, 1, 1, 1]], [[1 / .95, 0, 1 / .95, 1 / .95], [0, 1 / .95, 1 / .95, 0]],
                         [[0, 0, 1 / .9, 1 / .9], [1 / .9, 1 / .9, 0, 1 / .9]]], dtype=torch.float32)
    logits = m._cara_engine.forward(x.to(DEV), droppath=keep.to(DEV))
    with torch.no_grad():
        sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=3, drop_path_keep=keep, factored=True, bf16_sim=True)
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=3, drop_path_keep=keep)
    assert T.logits_ok(rel(logits, ref), rel(sim, ref))
    # a wrong mask moves the logits by far more than the rounding floor
    with torch.no_grad():
        wrong = O.vit_cara_forward(x, w, cp, s=0.1, depth=3)
    assert rel(wrong, ref) > 5 * rel(logits, ref)
    # engine-drawn masks: shape, values in {0, 1/keep}, block 0 never dropped
    dp = m._cara_engine.draw_droppath(m, 64, torch.device(DEV))
    assert dp.shape == (3, 2, 64) and torch.equal(dp[0], torch.ones(2, 64, device=DEV))
    assert all(min(abs(v), abs(v - 1 / 0.9)) < 1e-6 for v in torch.unique(dp[2]).tolist())
    torch.logsumexp(logits, 1).sum().backward()
    assert all(torch.isfinite(getattr(m, n).grad).all() for n in O.CP_NAMES)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_module_level_forwards_against_reference_vectors(precision):
    """The reference's patched Attention.forward / Mlp.forward (cara.py:15-60, :63-95) called on
    their own: golden case 2 of make_golden.py (block 5 of a depth-12 model, rank 8, x [2,7,768]) --
    outputs recorded from the reference's own modules; gradients against fp64 autograd of the
    oracle's as-written restatement.  Both operand builds (fp16: the per-op entry points of libcara_hip_f16.so, no loss scale on
    this path -- the gradient arriving from autograd is the caller's to scale, as with any fp16 module under torch.amp)."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case
    R, Lb, sb, sc, sx, sg = G["mod_cfg"].tolist()
    S = float(G["mod_scale"][0])
    w, cp = oracle_case(sg, sb, sc, R, 12, 32)
    m = build(w, cp, R, S, 12, 32, precision=precision).eval()
    bar_y, bar_gx, bar_gp = (1.5e-3, 3e-3, 5e-3) if precision == "fp16" else (1e-2, 2e-2, 3e-2)
    x = torch.randn(2, 7, 768, generator=torch.Generator().manual_seed(sx))
    blk = m.blocks[Lb]
    a_idx, a_aidx, m_idx = O.block_indices(12)[Lb]
    assert (blk.attn.idx, blk.attn.attn_idx, blk.mlp.idx) == (a_idx, a_aidx, m_idx)
    p = f"blocks.{Lb}."
    d = lambda t: t.double()  # noqa: E731
    for kind, mod, key in (("attn", blk.attn, "mod_attn_out"), ("mlp", blk.mlp, "mod_mlp_out")):
        xd = x.to(DEV).requires_grad_(True)
        for n in O.CP_NAMES:
            getattr(m, n).grad = None
        y = mod(xd)
        ref = torch.from_numpy(G[key])
        r = rel(y, ref)
        print(f"module {kind} [{precision}]: rel-L2 vs reference output {r:.2e}")
        assert r < bar_y, (kind, r)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        y.backward(gy.to(DEV))
        cpv = {k: d(v).clone().requires_grad_(True) for k, v in cp.items()}
        xr = d(x).clone().requires_grad_(True)
        if kind == "attn":
            yr = O.attn_as_written(xr, cpv, d(w[p + "attn.qkv.weight"]), d(w[p + "attn.qkv.bias"]), d(w[p + "attn.proj.weight"]),
                                   d(w[p + "attn.proj.bias"]), attn_idx=a_aidx, idx=a_idx, s=S, num_heads=12, scale=64 ** -0.5)
            touched = ("CP_A1", "CP_A2", "CP_A3", "CP_A4", "CP_R1", "CP_P1", "CP_P2", "CP_P3", "CP_R2", "CP_bias1")
        else:
            yr = O.mlp_as_written(xr, cpv, d(w[p + "mlp.fc1.weight"]), d(w[p + "mlp.fc1.bias"]), d(w[p + "mlp.fc2.weight"]),
                                  d(w[p + "mlp.fc2.bias"]), idx=m_idx, s=S)
            touched = ("CP_P1", "CP_P2", "CP_P3", "CP_R2", "CP_bias2", "CP_bias3")
        yr.backward(d(gy))
        assert rel(xd.grad, xr.grad) < bar_gx, (kind, rel(xd.grad, xr.grad))
        for n in touched:
            assert rel(getattr(m, n).grad, cpv[n].grad) < bar_gp, (kind, n, rel(getattr(m, n).grad, cpv[n].grad))


def test_blockwise_path_equals_fused_path():
    """Calling the blocks one by one (torch LayerNorm / residual around the module-level forwards)
    gives the fused whole-model logits up to bf16 rounding."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone(depth=2)
    cp = O.synthetic_cp(rank=16)
    x, _ = O.synthetic_batch(batch=2)
    m = build(w, cp, 16, 0.1, 2, 224).eval()
    with torch.no_grad():
        fused = m(x.to(DEV))
        pe = m.patch_embed.proj
        t = torch.nn.functional.conv2d(x.to(DEV), pe.weight, pe.bias, stride=16).flatten(2).transpose(1, 2)
        t = torch.cat((m.cls_token.expand(2, -1, -1), t), 1) + m.pos_embed
        for blk in m.blocks:
            t = blk(t)
        blockwise = m.head(m.norm(t)[:, 0])
    assert rel(blockwise, fused) < 1.2e-2 and torch.equal(blockwise.argmax(1), fused.argmax(1))


def test_recipe_fit_learns_and_keeps_reference_quirks():
    """cara_amd.recipe.fit = the loop of vit_cp.py:19-70 on a tiny synthetic task (depth 2)."""
    from cara_amd import cara, create_model
    from cara_amd.recipe import fit
    torch.manual_seed(0)
    m = cara({"model": create_model("vit_base_patch16_224_in21k", depth=2, num_classes=4, drop_path_rate=0.1), "rank": 8,
              "scale": 1.0, "l_mu": 1.0, "l_std": 0.0}).to(DEV)
    g = torch.Generator().manual_seed(1)
    y = torch.arange(16) % 4
    x = (torch.randn(16, 3, 224, 224, generator=g) * 0.3 + y.float().reshape(-1, 1, 1, 1)).to(DEV)   # class = mean level
    y = y.to(DEV)
    evals = []
    best, opt = fit(m, lambda epoch: [(x, y)], lambda: [(x, y)], epochs=21, lr=1e-2, on_eval=lambda e, a: evals.append((e, a)))
    assert [e for e, _ in evals] == [10, 20]                 # vit_cp.py:57
    assert not m.training                                     # vit_cp.py:75: eval() sticks
    assert best >= 0.75, evals                                # it learns the toy task
    assert all(p.grad is not None for n, p in m.named_parameters() if "CP" in n or "head" in n)
    assert all(p.grad is None for n, p in m.named_parameters() if not ("CP" in n or "head" in n))


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("rank,batch", [(8, 16), (64, 4), (32, 3)])
def test_baseline_configs_rank_variants(rank, batch, precision):
    """BASELINE.json configs[0] (rank 8, bs 16) and configs[3] (rank 64: Rp = 64 paths of the
    K-extension, skinny v1 and tskinny NT = 4) plus the reference's CLI default rank 32: depth-12
    ViT-B/16 logits and CP gradients against the oracle."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=batch)
    m = build(w, cp, rank, 0.1, 12, 224, precision=precision).eval()
    logits = m(x.to(DEV))
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, factored=True, bf16_sim=True) if precision == "bf16" else None
    check_logits(logits, ref, sim, precision, f"rank {rank} bs {batch}")
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1)
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    print(f"rank {rank} [{precision}]: worst CP-gradient rel-L2 {worst:.2e}")
    assert worst < grad_bar(precision)


# ---- the entry point bench.py times: CaraEngine.train_step ------------------------------------------------------
def _keep(depth, B, seed=11):
    """per-sample DropPath multipliers [depth, 2, B] as timm draws them (rates linspace(0, 0.1, depth)), fixed seed"""
    g = torch.Generator().manual_seed(seed)
    rates = torch.linspace(0, 0.1, depth)
    keep = (1 - rates).reshape(-1, 1, 1)
    return ((keep + torch.rand(depth, 2, B, generator=g)).floor() / keep).float()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_train_step_against_oracle(precision):
    """train_step(x, y, None): loss and p.grad of every trainable tensor (the views of the flat buffer) against fp32
    autograd of the as-written algorithm with the SAME DropPath masks; depth 3, batch 8 (1576 rows: the full-size
    GEMM kernels with the riding skinny products), train mode."""
    from oracle import cara_oracle as O
    depth, B = 3, 8
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=16)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, 16, 0.1, depth, 224, precision=precision).train()
    eng = m._cara_engine
    keep = _keep(depth, B)
    loss = eng.train_step(x.to(DEV), y.to(DEV), None, droppath=keep.to(DEV))
    cps = dict(cp)
    cps["CP_A1"], cps["CP_P1"] = cp["CP_A1"][:3 * depth], cp["CP_P1"][:9 * depth]
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    rloss, rlogits, gref = O.train_step_as_written(x, y, w, cps, head, s=0.1, depth=depth, drop_path_keep=keep)
    assert abs(loss.item() - rloss.item()) < (5e-4 if precision == "fp16" else 5e-3) * max(1.0, abs(rloss.item())), (loss.item(), rloss.item())
    if precision == "fp16":
        with torch.no_grad():
            check_logits(eng.forward(x.to(DEV), droppath=keep.to(DEV)), rlogits, None, precision, "train-mode forward, depth 3")
    worst = 0.0
    for n in O.CP_NAMES:
        p_ = getattr(m, n)
        assert p_.grad is not None and p_.grad.data_ptr() == eng._grad_views[n[3:]].data_ptr()      # views of ONE flat buffer
        worst = max(worst, rel(p_.grad, gref[n]))
    print(f"train_step [{precision}]: loss {loss.item():.5f} vs oracle {rloss.item():.5f}; worst CP-gradient rel-L2 {worst:.2e}")
    assert worst < grad_bar(precision)
    assert rel(m.head.weight.grad, gref["head.weight"]) < grad_bar(precision) and rel(m.head.bias.grad, gref["head.bias"]) < grad_bar(precision)
    # labels of the wrong dtype would make the cross-entropy kernel read out of bounds: refused
    from cara_amd._lib import CaraError
    with pytest.raises(CaraError):
        eng.train_step(x.to(DEV), y.to(DEV).int(), None, droppath=keep.to(DEV))


def test_three_adamw_steps_follow_the_oracle_trajectory():
    """Three train steps with AdamW (vit_cp.py:185 settings) on the device against the same three steps of the
    oracle on the CPU (fp32 autograd of the as-written algorithm + torch AdamW), same DropPath masks per step: the
    loss of every step and every trainable tensor afterwards.  Adam's first steps are lr * sign(g) for every
    element whatever |g| is, so elements whose gradient is below the bf16 noise can move the other way: the
    tensors are compared in rel-L2 (tight: an update is 1e-3 of a value) and the UPDATE by its direction."""
    from oracle import cara_oracle as O
    depth, B, steps = 3, 8, 3
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=16)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, 16, 0.1, depth, 224).train()
    eng = m._cara_engine
    trainable = eng.trainable_parameters()
    opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4)
    before = {n: getattr(m, n).detach().cpu().clone() for n in O.CP_NAMES}
    # oracle side: parameters as leaves, torch AdamW on the CPU
    cps = {k: torch.nn.Parameter(v.clone()) for k, v in cp.items()}
    cps["CP_A1"], cps["CP_P1"] = torch.nn.Parameter(cp["CP_A1"][:3 * depth].clone()), torch.nn.Parameter(cp["CP_P1"][:9 * depth].clone())
    hw, hb = torch.nn.Parameter(w["head.weight"].clone()), torch.nn.Parameter(w["head.bias"].clone())
    ropt = torch.optim.AdamW([cps[n] for n in O.CP_NAMES] + [hw, hb], lr=1e-3, weight_decay=1e-4)
    for it in range(steps):
        keep = _keep(depth, B, seed=100 + it)
        loss = eng.train_step(x.to(DEV), y.to(DEV), opt, droppath=keep.to(DEV))
        ww = dict(w)
        ww["head.weight"], ww["head.bias"] = hw, hb
        rl = torch.nn.functional.cross_entropy(O.vit_cara_forward(x, ww, cps, s=0.1, depth=depth, drop_path_keep=keep), y)
        ropt.zero_grad()
        rl.backward()
        ropt.step()
        print(f"step {it}: loss {loss.item():.5f} vs oracle {rl.item():.5f}")
        assert abs(loss.item() - rl.item()) < 5e-3 * max(1.0, abs(rl.item())), (it, loss.item(), rl.item())
    for n in O.CP_NAMES:
        dev_p, ref_p = getattr(m, n).detach().cpu(), cps[n].detach()
        # (the three bias vectors are ~0.02 in magnitude and move by lr = 1e-3 per step whatever |g| is: one element in twenty
        # stepping the other way once is already 5e-3 of the tensor -- measured 4.6e-3 ... 5.2e-3 on CP_bias2 from box to box)
        assert rel(dev_p, ref_p) < (8e-3 if "bias" in n else 5e-3), (n, rel(dev_p, ref_p))
        du, ru = (dev_p - before[n]).double().flatten(), (ref_p - before[n]).double().flatten()
        cos = (du @ ru / (du.norm() * ru.norm())).item()
        assert cos > 0.9, (n, cos)
    assert rel(m.head.weight.detach(), hw.detach()) < 5e-3


@pytest.mark.parametrize("rank,precision", [(16, "bf16"), (64, "bf16"), (64, "fp16")])   # (16, "fp16"): test_fp16_precision_at_the_headline_size
def test_headline_batch_64_whole_model(rank, precision):
    """BASELINE.json configs[1] (rank 16) and configs[3] (rank 64: Rp = 64 kernels -- adapter inside the N = 768 GEMMs,
    LayerNorm-fused contractions, riding products with 16 accumulator tiles) at their REAL size: ViT-B/16 depth 12,
    batch 64 -> M = 12 608 token rows (98.5 row tiles: the edge tile, 2.3 rounds of workgroups), through train_step.
    Logits of the same forward (eval path is bitwise the training forward) against the fp32 oracle and its bf16-rounded
    form; loss and every gradient of the train step against fp32 autograd of the as-written algorithm."""
    from oracle import cara_oracle as O
    B = 64
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, rank, 0.1, 12, 224, precision=precision).train()
    eng = m._cara_engine
    keep = _keep(12, B)
    loss = eng.train_step(x.to(DEV), y.to(DEV), None, droppath=keep.to(DEV))
    # expected values: the CPU oracle's as-written fp32 train step on these very inputs, computed once in the build
    # container (tests/golden/make_headline_fixtures.py: ~100 s of CPU per rank) and committed; recomputed here only if
    # the fixture is missing
    fx = os.path.join(os.path.dirname(__file__), "golden", f"headline_b64_r{rank}.npz")
    if os.path.exists(fx):
        F_ = np.load(fx)
        assert torch.equal(torch.from_numpy(F_["droppath"]), keep)          # same masks as the fixture was made with
        rloss, rlogits, sim = torch.tensor(float(F_["loss"])), torch.from_numpy(F_["logits"]), torch.from_numpy(F_["logits_bf16_sim"])
        gref = {k[len("grad_"):]: torch.from_numpy(F_[k]) for k in F_.files if k.startswith("grad_")}
    else:
        head = {"weight": w["head.weight"], "bias": w["head.bias"]}
        rloss, rlogits, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1, drop_path_keep=keep)
        with torch.no_grad():
            sim = O.vit_cara_forward(x, w, cp, s=0.1, drop_path_keep=keep, factored=True, bf16_sim=True)
    with torch.no_grad():
        logits = eng.forward(x.to(DEV), droppath=keep.to(DEV))
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    check_logits(logits, rlogits, sim, precision, f"batch 64, rank {rank}")
    differ = int((logits.argmax(1).cpu() != rlogits.argmax(1)).sum())
    print(f"batch 64, rank {rank} [{precision}]: loss {loss.item():.5f} vs {rloss.item():.5f}; worst CP-gradient rel-L2 {worst:.2e}; "
          f"class indices that differ from the fp32 oracle's, all {B} samples, no margin filter: {differ}")
    assert abs(loss.item() - rloss.item()) < (5e-4 if precision == "fp16" else 5e-3) * max(1.0, abs(rloss.item()))
    assert worst < grad_bar(precision) and rel(m.head.weight.grad, gref["head.weight"]) < grad_bar(precision)


def test_one_block_against_the_bf16_rounded_oracle():
    """"The kernels add no error of their own", asserted end to end on ONE block: the patched Attention.forward and
    Mlp.forward (skinny contraction + K-extension GEMM + fused attention + GELU epilogue, full-size kernels: 8 x 197
    rows) on bf16-representable inputs against the oracle's factored form rounded at the same points.  What is left
    is accumulation order and rounding ties: <= 1e-3 (measured 2.0e-4 / 3.2e-5)."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone(depth=2)
    cp = O.synthetic_cp(rank=16)
    m = build(w, cp, 16, 0.1, 2, 224).eval()
    cps = dict(cp)
    cps["CP_A1"], cps["CP_P1"] = cp["CP_A1"][:6], cp["CP_P1"][:18]
    fac = O.build_factored(cps, 0.1, depth=2)
    r = O.make_rounder(torch.bfloat16)
    xn = r(torch.randn(8, 197, 768, generator=torch.Generator().manual_seed(3)))
    blk, p = m.blocks[1], "blocks.1."
    with torch.no_grad():
        ya = blk.attn(xn.to(DEV))
        ym = blk.mlp(xn.to(DEV))
        ra = O._attn_factored(xn, w, p, fac[1], 12, 64 ** -0.5, r)
        rm = O._mlp_factored(xn, w, p, fac[1], r)
    ea, em = rel(ya, ra), rel(ym, rm)
    print(f"one block vs bf16-rounded oracle: attention {ea:.2e}, mlp {em:.2e}")
    assert ea < T.BLOCK_VS_SIM and em < T.BLOCK_VS_SIM, (ea, em)


# ---- drop-in on a foreign timm-shaped model, checkpoints ----------------------------------------------------------
def test_foreign_vit_class_runs_the_fused_path():
    """cara() on the ORACLE's VisionTransformer class (timm's names and attributes, another module): the adopted model's
    logits and CP gradients are bitwise those of this package's own container holding the same state dict; calling
    a block runs the patched module-level forwards."""
    from oracle import cara_oracle as O
    from cara_amd import cara
    w = O.synthetic_backbone(depth=3)
    cp = O.synthetic_cp(rank=16)
    x, y = O.synthetic_batch(batch=8)
    own = build(w, cp, 16, 0.1, 3, 224).eval()
    torch.manual_seed(0)
    foreign = cara({"model": O.create_vit("vit_base_patch16_224_in21k", depth=3, num_classes=100, drop_path_rate=0.1), "rank": 16,
                    "scale": 0.1, "l_mu": 1.5, "l_std": 0.1})
    assert type(foreign).__module__.startswith("oracle") and type(foreign.blocks[0].attn).__module__.startswith("oracle")
    missing, unexpected = foreign.load_state_dict(own.state_dict(), strict=True)
    foreign = foreign.to(DEV).eval()
    la, lb = own(x.to(DEV)), foreign(x.to(DEV))
    assert torch.equal(la, lb)
    torch.nn.functional.cross_entropy(la, y.to(DEV)).backward()
    ga = {n: getattr(own, n).grad.clone() for n in O.CP_NAMES}
    torch.nn.functional.cross_entropy(lb, y.to(DEV)).backward()
    assert all(torch.equal(ga[n], getattr(foreign, n).grad) for n in O.CP_NAMES)
    with torch.no_grad():
        t = torch.randn(2, 197, 768, device=DEV)
        assert torch.equal(foreign.blocks[1](t), own.blocks[1](t))      # eager Block.forward around the patched sub-modules


def test_npz_ingest_save_best_and_evaluate(tmp_path):
    """vit_cp.py:155,61-66,168-173 end to end: a JAX-layout .npz -> create_model(checkpoint_path=...) -> cara() ->
    the engine's bf16 HBM images; logits equal those of a twin that got the same tensors through load_state_dict;
    fit() writes the best-accuracy state dict under the reference's file name and replaces it on improvement;
    load_checkpoint() into a fresh model (--evaluate) reproduces the logits bitwise."""
    import numpy as np
    from cara_amd import cara, create_model
    from cara_amd.checkpoint import state_dict_to_jax
    from cara_amd.recipe import evaluate_only, fit
    torch.manual_seed(0)
    src = create_model("vit_base_patch16_224_in21k", depth=2, num_classes=21843)
    with torch.no_grad():
        for n, p_ in src.named_parameters():
            if "norm" not in n:
                p_.copy_(0.02 * torch.randn_like(p_))
    npz = str(tmp_path / "ViT-B_16.npz")
    np.savez(npz, **state_dict_to_jax(src))
    mk = lambda **kw: cara({"model": create_model("vit_base_patch16_224_in21k", depth=2, drop_path_rate=0.1, **kw), "rank": 8,  # noqa: E731
                            "scale": 1.0, "l_mu": 1.0, "l_std": 0.0})
    torch.manual_seed(1)
    a = mk(checkpoint_path=npz)
    a.reset_classifier(4)
    torch.manual_seed(1)
    b = mk()
    b.reset_classifier(4)
    b.load_state_dict({**src.state_dict(), **{k: v for k, v in a.state_dict().items() if k.startswith(("CP_", "head."))}})
    a, b = a.to(DEV), b.to(DEV)
    g = torch.Generator().manual_seed(1)
    y = torch.arange(16) % 4
    x = (torch.randn(16, 3, 224, 224, generator=g) * 0.3 + y.float().reshape(-1, 1, 1, 1)).to(DEV)
    y = y.to(DEV)
    a.eval(), b.eval()
    with torch.no_grad():
        assert torch.equal(a(x), b(x))
    sb = {"dataset": "toy", "seed": 7, "dir": str(tmp_path)}
    best, _ = fit(a, lambda epoch: [(x, y)], lambda: [(x, y)], epochs=21, lr=1e-2, save_best=sb)
    files = [f for f in os.listdir(tmp_path) if f.endswith(".pt")]
    assert len(files) == 1 and files[0] == os.path.basename(sb["path"]) and files[0] == f"vit_toy_{round(best, 5)}_seed_7.pt"
    with torch.no_grad():
        la = a(x)
    torch.manual_seed(5)
    c = mk()
    c.reset_classifier(4)
    c = c.to(DEV)
    acc = evaluate_only(c, sb["path"], [(x, y)])
    with torch.no_grad():
        lc = c(x)
    # the file holds the state of the BEST evaluation (epoch 10 or 20), a kept training after it only if 20 was best
    assert abs(acc - best) < 1e-9
    if best == (la.argmax(1) == y).float().mean().item():
        assert torch.equal(la, lc) or acc == best


def test_resident_split_on_the_device(tmp_path):
    """SURVEY 8f row 3 on the GPU: the VTAB file list decoded once into a device-resident tensor; an epoch's batches
    are the same images (same normalisation) as a CPU decode of the listed files, in the epoch permutation's order."""
    PIL = pytest.importorskip("PIL.Image")
    from cara_amd import data as D
    import numpy as np
    root = tmp_path / "toy"
    (root / "images").mkdir(parents=True)
    rng = np.random.default_rng(0)
    lines = []
    for i in range(10):
        arr = rng.integers(0, 255, size=(40 + i, 50, 3), dtype=np.uint8)
        PIL.fromarray(arr).save(root / "images" / f"im{i}.png")
        lines.append(f"images/im{i}.png {i % 3}")
    (root / "train800val200.txt").write_text("\n".join(lines))
    flist = str(root / "train800val200.txt")
    split = D.ResidentSplit(str(root), flist, device=DEV)
    cpu = D.ResidentSplit(str(root), flist, device="cpu")
    assert split.pixels.is_cuda and split.pixels.dtype == torch.uint8 and split.pixels.shape == (10, 3, 224, 224)
    want_all = torch.stack([D.decode_image(os.path.join(str(root), p_)) for p_, _ in cpu.imlist])      # the per-file CPU decode
    assert torch.allclose(split.images.cpu(), want_all, rtol=3e-7, atol=3e-7) and torch.equal(split.labels.cpu(), cpu.labels)
    got = list(split.train_batches(4, seed=0, rank=0, world=1)(3))
    want = list(cpu.train_batches(4, seed=0, rank=0, world=1)(3))
    assert len(got) == len(want) == 2
    for (xa, ya), (xb, yb) in zip(got, want):
        assert xa.is_cuda and xa.dtype == torch.float32 and torch.allclose(xa.cpu(), xb, rtol=3e-7, atol=3e-7) and torch.equal(ya.cpu(), yb)
    ev = list(split.eval_batches(8)())
    assert [b[0].shape[0] for b in ev] == [8, 2] and torch.allclose(torch.cat([b[0] for b in ev]).cpu(), want_all, rtol=3e-7, atol=3e-7)


def test_bench_self_launches_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with NO torchrun on the command line and no WORLD_SIZE in the environment: the
    script starts torch.distributed.run itself, as a child, before touching the GPU (rehearsal: both ranks on
    cuda:0 over gloo), and rank 0 prints the contract's JSON line with n_gpus = 2 and two ranks in the all-reduce."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CARA_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_allreduce"] == 2 and d["config"]["global_batch"] == 16
    assert d["roofline"]["launches_timed"] > 0 and len(d["roofline_top"]) == 3 and d["roofline_hbm"]


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's multi-rank plumbing on ONE GPU (CARA_BENCH_REHEARSAL=1: both ranks on cuda:0, gloo instead of RCCL):
    rendezvous, per-rank shards, the flat-gradient all-reduce inside train_step, max-over-ranks timing, one JSON line
    from rank 0 with the contract's keys.  Child processes, never an exec from this GPU-initialised one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CARA_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and "cpu_baseline" not in d          # the CPU baseline is a rank-0, N = 1 leg
    pm = d["precision_matched"]                                  # the fp16 (1e-3) build, timed the same way on every rank
    assert pm["dtype"] == "fp16" and pm["value"] > 0 and pm["roofline"]["bound"] == "mfma" and pm["steps_skipped_for_overflow"] == 0, pm


def test_bench_two_gpus_over_rccl_when_the_box_has_them():
    """The `nccl` (= RCCL) branch of bench.py under pytest on the first box that has two GPUs: process-group init with
    device_id, the HSA_ENABLE_IPC_MODE_LEGACY=0 assumption of the self-launch, one all-reduce per step over both ranks.
    Skipped on the one-GPU boxes of this pool (the gloo rehearsals above cover the plumbing there); no scaling number is
    asserted."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the RCCL branch needs two (covered over gloo by the rehearsal tests)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "CARA_BENCH_REHEARSAL")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "16"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_allreduce"] == 2 and d["config"]["backend"] == "rccl"
    assert d["config"]["global_batch"] == 32 and d["value"] > 0 and d["scaling"] == "weak"


# ---- SURVEY 8f row 1 on the device: a Flax-layout .npz, evaluated from the Flax layer definitions -------------------------
def _flax_dict(depth, grid, D=768, H=12, hd=64, P=16, classes=7, seed=0):
    """A hand-built dict in the published ViT-B_16.npz layout (conv kernel HWIO, per-head query/key/value kernels
    [D,H,hd] with biases [H,hd], out kernel [H,hd,D], Dense kernels [in,out]); magnitudes of a trained ViT."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda std, *s: (torch.randn(*s, generator=g, dtype=torch.float64) * std).numpy()  # noqa: E731
    w = {"embedding/kernel": rn(0.02, P, P, 3, D), "embedding/bias": rn(0.02, D), "cls": rn(0.02, 1, 1, D),
         "Transformer/posembed_input/pos_embedding": rn(0.02, 1, grid * grid + 1, D),
         "Transformer/encoder_norm/scale": 1.0 + rn(0.1, D), "Transformer/encoder_norm/bias": rn(0.02, D),
         "head/kernel": rn(0.02, D, classes), "head/bias": rn(0.02, classes)}
    for l in range(depth):
        pre = f"Transformer/encoderblock_{l}/"
        mha = pre + "MultiHeadDotProductAttention_1/"
        for ln in ("LayerNorm_0", "LayerNorm_2"):
            w[pre + ln + "/scale"], w[pre + ln + "/bias"] = 1.0 + rn(0.1, D), rn(0.02, D)
        w[pre + "MlpBlock_3/Dense_0/kernel"], w[pre + "MlpBlock_3/Dense_0/bias"] = rn(0.02, D, 4 * D), rn(0.02, 4 * D)
        w[pre + "MlpBlock_3/Dense_1/kernel"], w[pre + "MlpBlock_3/Dense_1/bias"] = rn(0.02, 4 * D, D), rn(0.02, D)
        w[mha + "out/kernel"], w[mha + "out/bias"] = rn(0.02, H, hd, D), rn(0.02, D)
        for n in ("query", "key", "value"):
            w[mha + n + "/kernel"], w[mha + n + "/bias"] = rn(0.02, D, H, hd), rn(0.02, H, hd)
    return w


def _flax_cara_forward(img, w, cp, s, depth, pos, H=12, hd=64, P=16):
    """ViT + CaRA evaluated in fp64 straight from the Flax layer definitions (einsums over the .npz arrays as they are
    stored) with the adapters of cara.py:26-42,50-58,72-82,87-93 added as written: independent of cara_amd.checkpoint,
    of the oracle's timm restatement and of the factored form."""
    T = lambda k: torch.from_numpy(np.asarray(w[k])).double()  # noqa: E731
    cpd = {k: v.double() for k, v in cp.items()}
    B, g = img.shape[0], img.shape[2] // P
    D = H * hd

    def ln(x, scale, bias):
        mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        return (x - mu) / torch.sqrt(var + 1e-6) * scale + bias
    nhwc = img.double().permute(0, 2, 3, 1).reshape(B, g, P, g, P, 3)
    x = (torch.einsum("bidjec,deco->bijo", nhwc, T("embedding/kernel")) + T("embedding/bias")).reshape(B, g * g, D)
    x = torch.cat([T("cls").expand(B, 1, D), x], 1) + pos.double()
    N = x.shape[1]
    for l in range(depth):
        pre = f"Transformer/encoderblock_{l}/"
        mha = pre + "MultiHeadDotProductAttention_1/"
        y = ln(x, T(pre + "LayerNorm_0/scale"), T(pre + "LayerNorm_0/bias"))
        dW = torch.einsum("r,kr,er,hr,dr->kehd", cpd["CP_R1"], cpd["CP_A1"][3 * l:3 * l + 3], cpd["CP_A2"], cpd["CP_A3"], cpd["CP_A4"])
        delta = torch.einsum("bne,kehd->kbhnd", y, dW)
        q, k, v = (torch.einsum("bne,ehd->bhnd", y, T(mha + n + "/kernel")) + T(mha + n + "/bias")[None, :, None, :] + s * delta[i]
                   for i, n in enumerate(("query", "key", "value")))
        a = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1) @ v
        ao = a.permute(0, 2, 1, 3).reshape(B, N, D)
        Tp = torch.einsum("r,jr,cr->jc", cpd["CP_R2"] * cpd["CP_P1"][9 * l], cpd["CP_P2"], cpd["CP_P3"])
        x = x + torch.einsum("bhnd,hdo->bno", a, T(mha + "out/kernel")) + T(mha + "out/bias") + s * (ao @ Tp.T + cpd["CP_bias1"])
        y = ln(x, T(pre + "LayerNorm_2/scale"), T(pre + "LayerNorm_2/bias"))
        Tu = torch.einsum("r,ar,jr,cr->ajc", cpd["CP_R2"], cpd["CP_P1"][9 * l + 1:9 * l + 5], cpd["CP_P2"], cpd["CP_P3"]).reshape(4 * D, D)
        up = y @ T(pre + "MlpBlock_3/Dense_0/kernel") + T(pre + "MlpBlock_3/Dense_0/bias") + s * (y @ Tu.T + cpd["CP_bias2"])
        h = torch.nn.functional.gelu(up)
        Td = torch.einsum("r,ar,jr,cr->ajc", cpd["CP_R2"], cpd["CP_P1"][9 * l + 5:9 * l + 9], cpd["CP_P2"], cpd["CP_P3"]).reshape(4 * D, D)
        x = x + h @ T(pre + "MlpBlock_3/Dense_1/kernel") + T(pre + "MlpBlock_3/Dense_1/bias") + s * (h @ Td + cpd["CP_bias3"])
    x = ln(x, T("Transformer/encoder_norm/scale"), T("Transformer/encoder_norm/bias"))
    return x[:, 0] @ T("head/kernel") + T("head/bias")


def _random_cp(depth, R, seed=5, D=768, H=12, hd=64):
    g = torch.Generator().manual_seed(seed)
    rn = lambda std, *s: torch.randn(*s, generator=g) * std  # noqa: E731
    return {"CP_A1": rn(0.3, 3 * depth, R), "CP_A2": rn(0.05, D, R), "CP_A3": rn(0.3, H, R), "CP_A4": rn(0.15, hd, R),
            "CP_P1": rn(0.3, 9 * depth, R), "CP_P2": rn(0.05, D, R), "CP_P3": rn(0.05, D, R),
            "CP_R1": 1.5 + rn(0.1, R), "CP_R2": 1.5 + rn(0.1, R),
            "CP_bias1": rn(0.02, D), "CP_bias2": rn(0.02, 4 * D), "CP_bias3": rn(0.02, D)}


def _bilinear_grid_resize(grid, new):
    """Independent restatement of a bilinear resize with half-pixel centres (align_corners = False, no antialiasing):
    numpy loops over the output grid.  grid [old, old, D] -> [new, new, D]."""
    old = grid.shape[0]
    out = np.zeros((new, new, grid.shape[2]), dtype=np.float64)
    sc = old / new

    def taps(i):
        src = max((i + 0.5) * sc - 0.5, 0.0)
        i0 = min(int(np.floor(src)), old - 1)
        i1 = min(i0 + 1, old - 1)
        return i0, i1, src - i0
    for i in range(new):
        a0, a1, fa = taps(i)
        for j in range(new):
            b0, b1, fb = taps(j)
            out[i, j] = (1 - fa) * ((1 - fb) * grid[a0, b0] + fb * grid[a0, b1]) + fa * ((1 - fb) * grid[a1, b0] + fb * grid[a1, b1])
    return out


def test_flax_layout_npz_through_the_device_path(tmp_path):
    """vit_cp.py:155 on the device, closed without the builder's own inverse mapping: a hand-built Flax-layout dict ->
    .npz -> create_model(checkpoint_path=...) -> cara() -> the HIP forward; logits against the Flax layer definitions
    evaluated directly (fp64 einsums over the arrays as stored) with the as-written adapters on top."""
    from cara_amd import cara, create_model
    depth, R, s = 2, 16, 0.1
    w = _flax_dict(depth, grid=14)
    npz = str(tmp_path / "ViT-B_16.npz")
    np.savez(npz, **w)
    torch.manual_seed(3)
    m = create_model("vit_base_patch16_224_in21k", checkpoint_path=npz, depth=depth, num_classes=7)
    m = cara({"model": m, "rank": R, "scale": s, "l_mu": 1.5, "l_std": 0.1})
    cp = _random_cp(depth, R)
    missing, unexpected = m.load_state_dict(cp, strict=False)
    assert not unexpected and not any(k.startswith("CP_") for k in missing)
    m = m.to(DEV).eval()
    img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        got = m(img.to(DEV))
    want = _flax_cara_forward(img, w, cp, s, depth, torch.from_numpy(w["Transformer/posembed_input/pos_embedding"]))
    base = _flax_cara_forward(img, w, {k: torch.zeros_like(v) for k, v in cp.items()}, s, depth,
                              torch.from_numpy(w["Transformer/posembed_input/pos_embedding"]))
    r = rel(got, want)
    moved = rel(base, want)
    print(f"\nflax-layout npz -> device: logits rel-L2 {r:.2e} vs the Flax definitions (the adapters move them by {moved:.2e})")
    assert moved > 5 * r                  # the adapters are live and far above the noise
    assert r <= T.LOGITS_ABS
    assert torch.equal(got.argmax(1).cpu(), want.argmax(1))


def test_flax_layout_npz_with_resized_position_embedding_at_384(tmp_path):
    """The 384-pixel path of timm's loader: the .npz holds a 14 x 14 (+ cls) position embedding, the model a 24 x 24 grid;
    checkpoint.resize_pos_embed against an INDEPENDENT bilinear restatement (numpy loops), then the whole thing through
    the device (577 tokens: the long-sequence attention kernels) against the Flax definitions with that embedding."""
    from cara_amd import cara, create_model
    depth, R, s = 1, 8, 0.1
    w = _flax_dict(depth, grid=14, seed=1)
    npz = str(tmp_path / "ViT-B_16.npz")
    np.savez(npz, **w)
    m = create_model("vit_base_patch16_224_in21k", checkpoint_path=npz, depth=depth, num_classes=7, img_size=384)
    pe = w["Transformer/posembed_input/pos_embedding"]
    grid = _bilinear_grid_resize(pe[0, 1:].reshape(14, 14, 768), 24).reshape(1, 576, 768)
    pos = torch.from_numpy(np.concatenate([pe[:, :1], grid], axis=1))
    assert m.pos_embed.shape == (1, 577, 768)
    assert torch.allclose(m.pos_embed.detach().double(), pos, atol=1e-6), (m.pos_embed.detach().double() - pos).abs().max()
    m = cara({"model": m, "rank": R, "scale": s, "l_mu": 1.5, "l_std": 0.1})
    cp = _random_cp(depth, R, seed=6)
    m.load_state_dict(cp, strict=False)
    m = m.to(DEV).eval()
    img = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        got = m(img.to(DEV))
    want = _flax_cara_forward(img, w, cp, s, depth, pos)
    r = rel(got, want)
    print(f"\nflax-layout npz @384 (resized position embedding) -> device: logits rel-L2 {r:.2e}")
    assert r <= T.LOGITS_ABS and torch.equal(got.argmax(1).cpu(), want.argmax(1))


def test_bf16x3_precision_mode_meets_the_1e3_logit_tolerance():
    """north_star: "within 1e-3 relative on bf16 logits".  The fast path sits at 5.5e-3 on this case because its MFMA
    operands carry 8 significant bits; cara_amd.precise.forward runs every product as three split-bf16
    MFMA products with fp32 accumulation and fp32 activations (cara_amd/precise.py) and must land inside the stated
    tolerance against the logits the REFERENCE's own cara.py produced (golden case 6: depth 2, 197 tokens, rank 16)."""
    from tests.golden.inputs import oracle_case
    R, depth, imgsz, sb, sc, sx, sg = G["d2_cfg"].tolist()
    w, cp = oracle_case(sg, sb, sc, R, depth, imgsz)
    m = build(w, cp, R, 0.1, depth, imgsz).eval()
    img = torch.randn(2, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(sx)).to(DEV)
    ref = torch.from_numpy(G["d2_logits"])
    from cara_amd import precise
    with torch.no_grad():
        fast = m(img)
        wide = precise.forward(m, img)
        wide9 = precise.forward(m, img.repeat(5, 1, 1, 1)[:9])      # more than eight images: sliced inside
    assert torch.equal(wide9[:2], wide) and wide9.shape[0] == 9
    r_fast, r_wide = rel(fast, ref), rel(wide, ref)
    print(f"\ndepth-2 golden logits vs the reference's fp32 output: bf16 fast path {r_fast:.2e}, bf16x3 {r_wide:.2e}")
    assert r_wide <= 1.0e-3, r_wide                    # north_star's number, met by construction
    assert r_wide < 0.1 * r_fast                       # ... and it is the operand width that does it
    assert torch.equal(wide.argmax(1).cpu(), ref.argmax(1))
    # it is an instrument, not a precision mode of cara() any more
    from cara_amd import cara, create_model
    from cara_amd._lib import CaraError
    with pytest.raises(CaraError, match="instrument"):
        cara({"model": create_model("vit_base_patch16_224_in21k", depth=1), "rank": 4, "scale": 0.1, "l_mu": 1.0, "l_std": 0.0, "precision": "bf16x3"})


def test_bf16x3_precision_mode_at_depth_12():
    """The same at the full depth (12 blocks, 197 tokens, rank 16, synthetic weights) against the fp32 as-written oracle: the
    fast path's 7e-3 is rounding that accumulates over the blocks; with split operands the logits stay inside 1e-3."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=16)
    x, _ = O.synthetic_batch(batch=2)
    m = build(w, cp, 16, 0.1, 12, 224).eval()
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1)
        fast = m(x.to(DEV))
        from cara_amd import precise
        wide = precise.forward(m, x.to(DEV))
    r_fast, r_wide = rel(fast, ref), rel(wide, ref)
    print(f"\ndepth 12 logits vs the fp32 oracle: bf16 fast path {r_fast:.2e}, bf16x3 {r_wide:.2e}")
    assert r_wide <= 1.0e-3 and r_wide < 0.1 * r_fast
    assert torch.equal(wide.argmax(1).cpu(), ref.argmax(1))


# ---- precision = "fp16": north_star's 1e-3 on the path that trains ---------------------------------------------------------
def test_fp16_precision_meets_the_1e3_logit_tolerance_on_the_reference_vectors():
    """precision = "fp16": the same HIP kernels compiled with IEEE-half MFMA operands (libcara_hip_f16.so: 11 significand bits at
    the bf16 MFMA rate, fp32 accumulation, fp32 residual stream) on golden case 6 -- depth 2, 197 tokens, rank 16: logits and all
    12 CP gradients the REFERENCE's own cara.py produced.  Forward inside 1e-3; the backward (static loss scale) with it."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case
    R, depth, imgsz, sb, sc, sx, sg = G["d2_cfg"].tolist()
    w, cp = oracle_case(sg, sb, sc, R, depth, imgsz)
    m = build(w, cp, R, 0.1, depth, imgsz).eval()
    img = torch.randn(2, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(sx)).to(DEV)
    ref = torch.from_numpy(G["d2_logits"])
    with torch.no_grad():
        fast = m(img)
    m._cara_engine.precision = "fp16"
    logits = m(img)
    r_fast, r_half = rel(fast, ref), rel(logits, ref)
    torch.logsumexp(logits, dim=1).sum().backward()
    worst = 0.0
    for n in O.CP_NAMES:
        g, gr = getattr(m, n).grad, torch.from_numpy(G["d2_grad_" + n])
        if n in ("CP_A1", "CP_P1"):
            gr = gr[:g.shape[0]]
        worst = max(worst, rel(g, gr))
    print(f"\ndepth-2 golden logits vs the reference's fp32 output: bf16 {r_fast:.2e}, fp16 {r_half:.2e}; fp16 worst CP-gradient rel-L2 {worst:.2e}")
    assert r_half <= FP16_LOGITS, r_half
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    assert worst < 0.2 * T.CP_GRAD, worst     # (bf16 measures 1e-2 here)


def test_fp16_precision_at_the_headline_size():
    """BASELINE.json configs[1] at its real size through train_step with precision = "fp16": ViT-B/16 depth 12, batch 64, rank 16,
    DropPath masks of the committed fixture -- logits against the fp32 as-written oracle (north_star's 1e-3), the class index of
    every one of the 64 samples (no margin filter), the loss and every gradient against fp32 autograd."""
    from oracle import cara_oracle as O
    B, rank = 64, 16
    fx = os.path.join(os.path.dirname(__file__), "golden", f"headline_b64_r{rank}.npz")
    if not os.path.exists(fx):
        pytest.skip("the committed headline fixture is missing")
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, rank, 0.1, 12, 224).train()
    eng = m._cara_engine
    eng.precision = "fp16"
    keep = _keep(12, B)
    F_ = np.load(fx)
    assert torch.equal(torch.from_numpy(F_["droppath"]), keep)
    rloss, rlogits = float(F_["loss"]), torch.from_numpy(F_["logits"])
    gref = {k[len("grad_"):]: torch.from_numpy(F_[k]) for k in F_.files if k.startswith("grad_")}
    loss = eng.train_step(x.to(DEV), y.to(DEV), None, droppath=keep.to(DEV))
    with torch.no_grad():
        logits = eng.forward(x.to(DEV), droppath=keep.to(DEV))
    r_ref = rel(logits, rlogits)
    differ = int((logits.argmax(1).cpu() != rlogits.argmax(1)).sum())
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    print(f"\nfp16, batch 64, rank 16: logits rel-L2 vs fp32 oracle {r_ref:.2e}; class indices that differ: {differ} of {B}; "
          f"loss {loss.item():.5f} vs {rloss:.5f}; worst CP-gradient rel-L2 {worst:.2e}; head {rel(m.head.weight.grad, gref['head.weight']):.2e}")
    assert r_ref <= FP16_LOGITS, r_ref
    assert differ == 0
    assert abs(loss.item() - rloss) < 5e-4 * max(1.0, abs(rloss))
    assert worst < 0.2 * T.CP_GRAD and rel(m.head.weight.grad, gref["head.weight"]) < 0.2 * T.CP_GRAD
    assert all(torch.isfinite(getattr(m, n).grad).all() for n in O.CP_NAMES)
    # three AdamW steps stay finite and the loss moves (the loss scale is divided out before the optimiser sees the gradients)
    from cara_amd.optim import AdamW
    opt = AdamW(eng.trainable_parameters(), lr=1e-3, weight_decay=1e-4)
    losses = [eng.train_step(x.to(DEV), y.to(DEV), opt, droppath=keep.to(DEV)).item() for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_fp16_precision_refuses_what_it_does_not_run():
    from oracle import cara_oracle as O
    from cara_amd._lib import CaraError
    w = O.synthetic_backbone(depth=1)
    cp = O.synthetic_cp(rank=8, depth=1)
    m = build(w, cp, 8, 0.1, 1, 224).eval()
    m._cara_engine.precision = "fp16"
    assert torch.isfinite(m.blocks[0].attn(torch.zeros(1, 197, 768, device=DEV))).all()   # (module-level entries run in fp16 since r05)
    m._cara_engine.weight_dropout = "exact"
    with pytest.raises(CaraError):
        m.train()(torch.zeros(1, 3, 224, 224, device=DEV))


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("rank,img,batch", [(48, 160, 3), (5, 96, 1), (33, 224, 2)])
def test_odd_ranks_token_counts_and_batches(rank, img, batch, precision):
    """Shapes off the beaten path, whole model (depth 2) against the oracle: ranks that are no multiple of 16 and cross the
    Rp = 32 / 64 boundary (5, 33, 48), token counts other than 197 (101 at 160 px, 37 at 96 px: the short-sequence attention
    kernels, ragged row tiles everywhere), batch 1."""
    from oracle import cara_oracle as O
    depth = 2
    w = O.synthetic_backbone(depth=depth, img=img)
    cp = O.synthetic_cp(rank=rank, depth=depth)
    x, y = O.synthetic_batch(batch=batch, img=img)
    m = build(w, cp, rank, 0.1, depth, img, precision=precision).eval()
    logits = m(x.to(DEV))
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=depth)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=depth, factored=True, bf16_sim=True)
    check_logits(logits, ref, sim, precision, f"rank {rank}, {img} px ({(img // 16) ** 2 + 1} tokens), batch {batch}")
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1, depth=depth)
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    print(f"rank {rank}, {img} px [{precision}]: worst CP gradient {worst:.2e}")
    assert worst < grad_bar(precision), worst


def test_forward_and_backward_capture_into_a_hip_graph():
    """include/cara_hip.h promises that cara_vit_forward / cara_vit_backward only enqueue work on the caller's stream (no
    allocation, no synchronisation, no state): the pair must therefore capture into a hipGraph, and replaying the graph must
    reproduce the eager results bitwise -- also after the parameters changed in place between replays."""
    from oracle import cara_oracle as O
    depth, B = 3, 4
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=16, depth=depth)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, 16, 0.1, depth, 224).train()
    eng = m._cara_engine
    keep = _keep(depth, B).to(DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    eng.train_step(xd, yd, None, droppath=keep)                 # eager: ingests the weights, sizes workspace and grad buffers
    want_loss = eng.train_step(xd, yd, None, droppath=keep).clone()
    want = {n: getattr(m, n).grad.clone() for n in O.CP_NAMES}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            loss = eng.train_step(xd, yd, None, droppath=keep)
    torch.cuda.current_stream().wait_stream(side)
    for n in O.CP_NAMES:
        getattr(m, n).grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss, want_loss)
    assert all(torch.equal(getattr(m, n).grad, want[n]) for n in O.CP_NAMES)
    # parameters move in place (what an optimizer step does): the replay sees them
    with torch.no_grad():
        m.CP_P2.mul_(1.5)
    g.replay()
    torch.cuda.synchronize()
    eager = eng.train_step(xd, yd, None, droppath=keep)
    torch.cuda.synchronize()
    assert torch.equal(loss, eager) and not torch.equal(loss, want_loss)


@pytest.mark.parametrize("switch", ["CARA_EPI_RIDERS=1", "CARA_FC1_SIDE=1", "CARA_DV=3"])
def test_rider_placement_switches_pass_the_whole_model_parity_tests(switch):
    """Three measured-and-off placements of the backward's heavy riders (read once per process): CARA_EPI_RIDERS=1 -- fc1's dVs / dc and
    fc2's dU out of the fc2 dX epilogue, gelu'(u) kept as IEEE half by fc1 forward; CARA_FC1_SIDE=1 -- the same products as a launch on a
    side stream under the fc1 dX GEMM; CARA_DV=3 -- fc1's and qkv's dVs / dc out of the A tiles of their own dX GEMM.  The batch-64
    whole-model test, the train-step test and the fp16 headline test run again in a process with the switch set."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **dict([switch.split("=")]))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_model_gpu.py"), "-x", "-q", "-s", "-k",
                          "headline_batch_64_whole_model or train_step_against_oracle or fp16_precision_at_the_headline_size"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join(out.stdout.strip().split("\n")[-12:])
    print(tail)
    assert out.returncode == 0, tail + out.stderr[-2000:]
    assert "passed" in tail and "failed" not in tail


def test_fp16_overflow_skips_the_step_and_backs_the_scale_off():
    """an inf / NaN produced in half range must not reach AdamW's moments.  A loss scale far too large for these
    gradients (2^30) overflows the 16-bit dY of the first layers: the kernels that write the final gradients raise the found-inf
    word, cara_amd.optim.AdamW's launch changes nothing, cara_amp_update halves the scale and counts the skip -- all on the
    device.  The following steps (the scale halving each time until the pass is finite) then train normally."""
    from oracle import cara_oracle as O
    from cara_amd.optim import AdamW
    depth, B = 2, 4
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=16, depth=depth)
    x, y = O.synthetic_batch(batch=B)
    m = build(w, cp, 16, 0.1, depth, 224).train()
    eng = m._cara_engine
    eng.precision = "fp16"
    opt = AdamW(eng.trainable_parameters(), lr=1e-3, weight_decay=1e-4)
    keep = _keep(depth, B).to(DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    eng.train_step(xd, yd, opt, droppath=keep)                    # a clean step at the default scale
    assert eng.skipped_steps == 0 and eng.loss_scale == eng.FP16_LOSS_SCALE
    eng._amp(xd.device)[0] = 2.0 ** 30
    before = {n: getattr(m, n).detach().clone() for n in O.CP_NAMES}
    mom = {n: opt.state[getattr(m, n)]["exp_avg"].clone() for n in O.CP_NAMES}
    loss = eng.train_step(xd, yd, opt, droppath=keep)
    assert torch.isfinite(loss)                                   # the loss itself is unscaled
    assert eng.skipped_steps == 1 and eng.loss_scale == 2.0 ** 29
    assert all(torch.equal(getattr(m, n).detach(), before[n]) for n in O.CP_NAMES)
    assert all(torch.equal(opt.state[getattr(m, n)]["exp_avg"], mom[n]) for n in O.CP_NAMES)
    # a foreign optimiser is stepped behind a host-side look at the word: skipped as well
    topt = torch.optim.AdamW(eng.trainable_parameters(), lr=1e-3)
    eng.train_step(xd, yd, topt, droppath=keep)
    assert eng.skipped_steps == 2 and all(torch.equal(getattr(m, n).detach(), before[n]) for n in O.CP_NAMES)
    # the scale keeps halving until the pass is finite; then the parameters move again and stay finite
    for _ in range(24):
        eng.train_step(xd, yd, opt, droppath=keep)
    assert eng.loss_scale < 2.0 ** 20
    assert any(not torch.equal(getattr(m, n).detach(), before[n]) for n in O.CP_NAMES)
    assert all(torch.isfinite(getattr(m, n)).all() for n in O.CP_NAMES)
    # gradients under the (now arbitrary) scale are the unscaled ones: against fp32 autograd
    eng.train_step(xd, yd, None, droppath=keep)
    cps = {k: getattr(m, k).detach().cpu() for k in O.CP_NAMES}
    head = {"weight": m.head.weight.detach().cpu(), "bias": m.head.bias.detach().cpu()}
    _, _, gref = O.train_step_as_written(x, y, w, cps, head, s=0.1, depth=depth, drop_path_keep=keep.cpu())
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    assert worst < 0.2 * T.CP_GRAD, worst


def test_graphed_train_step_follows_the_eager_one():
    """bench.py --graph / recipe.fit(graph=True): the whole step -- forward, cross-entropy, backward, AdamW with its step count and
    learning rate in device memory -- replayed from a hipGraph.  Five steps with a learning rate that changes every step and batches
    that alternate: (i) replayed vs the SAME capturable optimiser issued eagerly: bitwise (losses and every CP tensor) -- the capture
    changes nothing; (ii) the capturable optimiser vs the ordinary one: its bias corrections are fp32 powf in the kernel instead of
    the host's doubles, which moves parameters by one ulp per step -- and a bf16 pipeline answers a one-ulp change of every parameter
    with ~1e-5 of the loss and ~1e-3 of the gradients (measured: tools, r05), so that comparison is held to 1e-3 / 3e-4.  In bf16
    and -- with the loss-scale kernels in the graph -- in fp16; fit(graph=True) runs end to end."""
    from oracle import cara_oracle as O
    from cara_amd.optim import AdamW
    from cara_amd.recipe import GraphedTrainStep, fit
    depth, B = 2, 4
    w = O.synthetic_backbone(depth=depth)
    cp = O.synthetic_cp(rank=16, depth=depth)
    xa, ya = O.synthetic_batch(batch=B)
    xb, yb = O.synthetic_batch(batch=B, seed_x=5, seed_y=6)
    batches = [(xa.to(DEV), ya.to(DEV)), (xb.to(DEV), yb.to(DEV))]
    for precision in PRECISIONS:
        out = {}
        for mode in ("eager", "capturable", "graph"):
            m = build(w, cp, 16, 0.1, depth, 224, drop_path_rate=0.0, precision=precision).train()
            eng = m._cara_engine
            opt = AdamW(eng.trainable_parameters(), lr=1e-3, weight_decay=1e-4, capturable=(mode != "eager"))
            gstep = GraphedTrainStep(eng, opt) if mode == "graph" else None
            losses = []
            for it in range(5):
                opt.param_groups[0]["lr"] = 1e-3 * (0.7 ** it)
                x, y = batches[it % 2]
                if mode == "graph":
                    loss = gstep(x, y)
                else:
                    if mode == "capturable":
                        opt.advance()
                    loss = eng.train_step(x, y, opt)
                losses.append(loss.item())
            out[mode] = (losses, {n: getattr(m, n).detach().clone() for n in O.CP_NAMES})
        assert out["graph"][0] == out["capturable"][0], (precision, out["graph"][0], out["capturable"][0])
        assert all(torch.equal(out["graph"][1][n], out["capturable"][1][n]) for n in O.CP_NAMES), precision
        assert np.allclose(out["eager"][0], out["graph"][0], rtol=1e-3), (precision, out["eager"][0], out["graph"][0])
        for n in O.CP_NAMES:   # (the bias tensors' updates are the smallest against their values: 3.1e-4 measured on one build, r05)
            assert rel(out["graph"][1][n], out["eager"][1][n]) < (6e-4 if "bias" in n else 3e-4), (precision, n)
        assert out["eager"][0][-1] < out["eager"][0][0]
    m = build(w, cp, 16, 0.1, depth, 224).train()
    best, _ = fit(m, lambda epoch: batches, lambda: batches[:1], epochs=11, lr=1e-2, graph=True)
    assert 0.0 <= best <= 1.0 and not m.training      # (evaluated at epoch 10; the eval mode sticks and the graph was re-captured for it)
