"""GPU parity of the whole adapted ViT (cara_vit_forward / cara_vit_backward through the Python
mirror of the reference API) against (a) vectors recorded from the reference's own cara.py and
(b) the CPU oracle on the same seeded inputs.

Tolerances.  The device path rounds GEMM operands and stored activations to bf16 (fp32
accumulate, fp32 residual stream, fp32 softmax/LayerNorm statistics); the reference is fp32.
BASELINE.json asks for 1e-3 relative on bf16 logits.  With 8-bit-mantissa operands that is below
the rounding floor: the oracle evaluated with the SAME bf16 rounding points (``bf16_sim``) is
itself 6.6e-3 (depth 2, golden case) away from the fp32 reference, and two implementations that
are not bitwise identical decorrelate within a few rounding stages, so they sit ~one floor apart
too.  What is asserted is therefore: (1) device-vs-fp32-reference error <= 1.5 x the error the
rounding model predicts (the kernels add no error of their own), (2) an absolute cap of 1.5e-2
(SURVEY.md section 7 H3 measured 9.5e-3 for plain autocast), (3) exact class indices wherever
the fp32 top-2 margin exceeds the noise.  Measured values are printed (-s) and recorded in
DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cara_reference_vectors.npz"))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def build(w, cp, rank, scale, depth, img, num_classes=100, drop_path_rate=0.1, name="vit_base_patch16_224_in21k", cp_length=4):
    from cara_amd import cara, create_model
    m = create_model(name, drop_path_rate=drop_path_rate, depth=depth, img_size=img, num_classes=num_classes)
    m = cara({"model": m, "rank": rank, "scale": scale, "l_mu": 1.5, "l_std": 0.1, "cp_length": cp_length})
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
    assert r_ref < 1.5e-2 and r_ref < 1.5 * max(r_model, 4e-3), (r_ref, r_model)
    assert r_sim < 1.5 * max(r_model, 4e-3), (r_sim, r_model)
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
        assert r < 3e-2, (n, r)     # bf16 activations/gradients vs fp32 autograd of the dense form
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
    assert r < 1.5e-2 and rel(rlogits, nodrop) > 5 * r and rel(other, rlogits) > 5 * r
    assert abs(loss.item() - rloss.item()) < 2e-2 * max(1.0, abs(rloss.item()))
    worst = 0.0
    for n in O.CP_NAMES:
        rr = rel(getattr(m, n).grad, gref[n])
        worst = max(worst, rr)
        assert rr < 4e-2, (n, rr)
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
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16, factored=True, bf16_sim=True)
    r_ref, r_sim, r_model = rel(logits, ref), rel(logits, sim), rel(sim, ref)
    print(f"ViT-L/16@384 logits rel-L2: vs fp32 oracle {r_ref:.2e}, vs bf16-rounded oracle {r_sim:.2e} (rounding model {r_model:.2e})")
    assert r_ref < 2e-2 and r_ref < 1.5 * max(r_model, 4e-3), (r_ref, r_model)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * (logits.cpu() - ref).abs().max()
    assert torch.equal(logits.argmax(1).cpu()[safe], ref.argmax(1)[safe])
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1, depth=24, num_heads=16)
    worst = 0.0
    for n in O.CP_NAMES:
        r = rel(getattr(m, n).grad, gref[n])
        worst = max(worst, r)
        assert r < 6e-2, (n, r)
    print(f"ViT-L/16@384 worst CP-gradient rel-L2 vs fp32 autograd of the as-written form: {worst:.2e}")
    assert rel(m.head.weight.grad, gref["head.weight"]) < 3e-2


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
    assert r_ref < 1.5e-2 and r_ref < 1.5 * max(r_model, 4e-3), (r_ref, r_model)
    assert r_sim < 1.5 * max(r_model, 4e-3), (r_sim, r_model)
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
        assert r < 5e-2, (n, r)
    assert rel(m.head.weight.grad, gref["head.weight"]) < 2e-2
    assert rel(m.head.bias.grad, gref["head.bias"]) < 2e-2


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
    assert e < 1.5 * max(rel(sim, rlogits), 4e-3) and e < 1.5e-2
    for k in cpo:
        gk = getattr(m, k).grad
        assert gk is not None and torch.isfinite(gk).all(), k
        if rg[k].norm() > 0:
            ek = rel(gk, rg[k])
            print(f"  d{k}: rel {ek:.2e}")
            assert ek < 6e-2, (k, ek)


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
    assert rel(logits, ref) < 1.5 * max(rel(sim, ref), 4e-3) and rel(logits, ref) < 1.5e-2
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


def test_gradients_finite_difference_direction():
    """Directional derivative of the loss along a random CP direction vs the analytic gradient
    (device path only; catches sign/scale errors independently of the oracle)."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone(depth=2)
    cp = O.synthetic_cp(rank=16)
    x, y = O.synthetic_batch(batch=16)   # the mean over more samples averages the forward noise down
    m = build(w, cp, 16, 1.0, 2, 224).eval()
    xd, yd = x.to(DEV), y.to(DEV)
    loss = torch.nn.functional.cross_entropy(m(xd), yd)
    loss.backward()
    g = torch.Generator().manual_seed(5)
    num, ana = 0.0, 0.0
    dirs = {n: torch.randn(getattr(m, n).shape, generator=g).to(DEV) for n in ("CP_A2", "CP_P2", "CP_P1", "CP_R2")}
    ana = sum((getattr(m, n).grad * d).sum().item() for n, d in dirs.items())
    def central(eps):
        vals = []
        for sgn in (1, -1):
            with torch.no_grad():
                for n, d in dirs.items():
                    getattr(m, n).add_(sgn * eps * d)
                vals.append(torch.nn.functional.cross_entropy(m(xd), yd).item())
                for n, d in dirs.items():
                    getattr(m, n).sub_(sgn * eps * d)
        return (vals[0] - vals[1]) / (2 * eps)

    # A coarse check by construction: along a random direction of this size (s = 1) the loss is strongly curved
    # (the central difference moves by 15 % between eps = 2e-2 and 4e-2) and below 1e-2 the bf16 forward noise
    # (~1e-3 per loss value) takes over, so no step size gives better than ~10 %.  It catches what it is for --
    # sign and factor-of-two errors -- independently of the oracle; the tight check of every CP gradient is the
    # comparison with fp32 autograd of the as-written algorithm in the tests above.
    d4, d2 = central(4e-2), central(2e-2)
    print(f"directional derivative: central differences {d4:.4f} (eps 4e-2), {d2:.4f} (2e-2); analytic {ana:.4f}")
    assert abs(d2 - ana) <= 0.25 * abs(ana) + 1e-3 and abs(d4 - ana) <= 0.35 * abs(ana) + 1e-3, (d4, d2, ana)


def test_module_level_forwards_against_reference_vectors():
    """The reference's patched Attention.forward / Mlp.forward (cara.py:15-60, :63-95) called on
    their own: golden case 2 of make_golden.py (block 5 of a depth-12 model, rank 8, x [2,7,768]) --
    outputs recorded from the reference's own modules; gradients against fp64 autograd of the
    oracle's as-written restatement."""
    from oracle import cara_oracle as O
    from tests.golden.inputs import oracle_case
    R, Lb, sb, sc, sx, sg = G["mod_cfg"].tolist()
    S = float(G["mod_scale"][0])
    w, cp = oracle_case(sg, sb, sc, R, 12, 32)
    m = build(w, cp, R, S, 12, 32).eval()
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
        print(f"module {kind}: rel-L2 vs reference output {r:.2e}")
        assert r < 1e-2, (kind, r)
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
        assert rel(xd.grad, xr.grad) < 2e-2, (kind, rel(xd.grad, xr.grad))
        for n in touched:
            assert rel(getattr(m, n).grad, cpv[n].grad) < 3e-2, (kind, n, rel(getattr(m, n).grad, cpv[n].grad))


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


@pytest.mark.parametrize("rank,batch", [(8, 16), (64, 4), (32, 3)])
def test_baseline_configs_rank_variants(rank, batch):
    """BASELINE.json configs[0] (rank 8, bs 16) and configs[3] (rank 64: Rp = 64 paths of the
    K-extension, skinny v1 and tskinny NT = 4) plus the reference's CLI default rank 32: depth-12
    ViT-B/16 logits and CP gradients against the oracle."""
    from oracle import cara_oracle as O
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=batch)
    m = build(w, cp, rank, 0.1, 12, 224).eval()
    logits = m(x.to(DEV))
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, factored=True, bf16_sim=True)
    r_ref, r_model = rel(logits, ref), rel(sim, ref)
    print(f"rank {rank} bs {batch}: logits rel-L2 vs fp32 oracle {r_ref:.2e} (rounding model {r_model:.2e})")
    assert r_ref < 1.5e-2 and r_ref < 1.5 * max(r_model, 4e-3)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * (logits.cpu() - ref).abs().max()
    assert torch.equal(logits.argmax(1).cpu()[safe], ref.argmax(1)[safe])
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, gref = O.train_step_as_written(x, y, w, cp, head, s=0.1)
    worst = max(rel(getattr(m, n).grad, gref[n]) for n in O.CP_NAMES)
    print(f"rank {rank}: worst CP-gradient rel-L2 {worst:.2e}")
    assert worst < 6e-2


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
