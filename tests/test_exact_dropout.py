"""Exact weight-space dropout mode: the mask hash (CPU mirror == library), the merged weights, the gradient
contraction and the column sums against float64 references; whole-model parity lives in test_model_gpu.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from cara_amd import _lib as L
from cara_amd.dropout import keep_mask, keep_hash_np

DEV = "cuda"


def test_hash_mirror_matches_library():
    lib = L.lib()
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 2 ** 32, size=2000, dtype=np.uint64).astype(np.uint32)
    for seed, lin in ((0, 0), (12345, 7), (2 ** 32 - 1, 47), (99, 1000)):
        ref = np.array([lib.cara_weight_dropout_hash(int(i), seed, lin) for i in idx[:300]], dtype=np.uint32)
        assert np.array_equal(keep_hash_np(idx[:300], seed, lin), ref)
    m = keep_mask(768, 3072, 0.1, seed=3, linear_id=5)
    assert m.shape == (768, 3072) and abs(float(m.mean()) - 0.9) < 2e-3          # keep probability 1 - p
    assert not np.array_equal(m, keep_mask(768, 3072, 0.1, seed=4, linear_id=5))
    assert keep_mask(5, 7, 0.0, seed=1, linear_id=0).all()


def _rnd(*shape, seed, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dtype).to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("out,inn,Rp,p", [(768, 768, 32, 0.1), (3072, 768, 32, 0.1), (768, 3072, 64, 0.25), (100, 70, 32, 0.0)])
def test_materialize_merge(out, inn, Rp, p):
    lib = L.lib()
    W = _rnd(out, inn, seed=1, scale=0.02)
    U, Vs = _rnd(inn, Rp, seed=2, scale=0.2), _rnd(out, Rp, seed=3, scale=0.05)
    Weff = torch.full((out, inn), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.check(lib.cara_materialize_merge(L.ptr(W), L.ptr(U), L.ptr(Vs), Rp, out, inn, C.c_float(p), 11, 5, L.ptr(Weff), L.stream()), "merge")
    keep = torch.from_numpy(keep_mask(out, inn, p, seed=11, linear_id=5)).to(DEV)
    ref = W.double() + keep.double() / (1.0 - p) * (Vs.double() @ U.double().t())
    err = (Weff.double() - ref).abs()
    assert (err <= 2 ** -8 * ref.abs() + 1e-6).all(), float(err.max())
    if p > 0:   # dropped elements are exactly the frozen weight
        assert torch.equal(Weff[~keep], W[~keep])
    # W == NULL: the masked delta alone, at ITS OWN precision (the B3 operand of the exact mode's GEMMs)
    Dm = torch.full((out, inn), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.check(lib.cara_materialize_merge(None, L.ptr(U), L.ptr(Vs), Rp, out, inn, C.c_float(p), 11, 5, L.ptr(Dm), L.stream()), "delta")
    dref = keep.double() / (1.0 - p) * (Vs.double() @ U.double().t())
    assert ((Dm.double() - dref).abs() <= 2 ** -8 * dref.abs() + 1e-6).all()      # (fp32 sum of 32..64 products, then one bf16 rounding)
    if p > 0:
        assert torch.count_nonzero(Dm[~keep]) == 0


@pytest.mark.gpu
def test_tiny_adapter_survives_next_to_the_frozen_weight():
    """The reference zero-initialises A2 / P2, so for a long stretch of training the materialised adapter is 1e-8 .. 1e-6
    next to weights of ~2e-2: merged into ONE bf16 weight (half an ulp of 0.02 is 6e-5) it would be rounded away
    completely and the train-mode forward would be the frozen backbone's.  The exact mode therefore runs
    y = x W^T + x Dm^T as two products accumulated in fp32 (cara_gemm_args::B3): the adapter's contribution comes out
    at the precision of its OWN bf16 image."""
    M, N, K = 1500, 768, 768
    X = _rnd(M, K, seed=1)
    W = _rnd(N, K, seed=2, scale=0.02)
    D = _rnd(N, K, seed=3, scale=1e-6)                      # |delta| ~ 1e-6: far below W's ulp
    merged = (W.float() + D.float()).to(torch.bfloat16)
    assert (merged == W).float().mean() > 0.95               # what a pre-merged weight keeps of it: (almost) nothing
    base = torch.empty(M, N, dtype=torch.float32, device=DEV)
    both = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.gemm(X, W, base, epi=L.EPI_F32)
    L.gemm(X, W, both, epi=L.EPI_F32, B3=D)
    contrib = (both.double() - base.double())
    ref = X.double() @ D.double().t()
    # fp32 accumulation next to an O(1) sum: absolute noise ~1e-7 on a contribution of ~3e-5
    rel = ((contrib - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel
    full = X.double() @ (W.double() + D.double()).t()
    assert ((both.double() - full).norm() / full.norm()).item() < 1e-5
    # ragged shapes, GELU epilogue, K-extension together with B3
    M, N, K = 333, 300, 128
    X, W, D = _rnd(M, K, seed=4), _rnd(N, K, seed=5, scale=0.05), _rnd(N, K, seed=6, scale=0.01)
    A2, B2, bias = _rnd(M, 32, seed=7), _rnd(N, 32, seed=8, scale=0.05), _rnd(N, seed=9, dtype=torch.float32)
    h, u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.gemm(X, W, h, epi=L.EPI_GELU, C2=u, B3=D, A2=A2, B2=B2, bias=bias)
    ref = X.double() @ (W.double() + D.double()).t() + A2.double() @ B2.double().t() + bias.double()
    assert ((u.double() - ref).abs() <= 2 ** -7 * ref.abs() + 5e-3).all()
    assert ((h.double() - torch.nn.functional.gelu(ref)).abs() <= 2 ** -7 * ref.abs() + 5e-3).all()


@pytest.mark.gpu
@pytest.mark.parametrize("out,inn,Rp,p", [(768, 768, 32, 0.1), (3072, 768, 32, 0.1), (768, 3072, 64, 0.25), (100, 70, 32, 0.0)])
def test_dropout_grad_contract_and_colsum(out, inn, Rp, p):
    lib = L.lib()
    dW = _rnd(out, inn, seed=1, dtype=torch.float32)
    U, Vs = _rnd(inn, Rp, seed=2, scale=0.2), _rnd(out, Rp, seed=3, scale=0.05)
    dU = torch.full((inn, Rp), float("nan"), device=DEV)
    dVs = torch.full((out, Rp), float("nan"), device=DEV)
    # the dense gradient arrives as split-K slabs that the kernels sum: hand it over in three pieces
    slabs = torch.stack([0.5 * dW, 0.25 * dW, 0.25 * dW]).contiguous()
    scratch = torch.empty(lib.cara_dropout_grad_scratch_bytes(out, inn, Rp), dtype=torch.uint8, device=DEV)
    L.check(lib.cara_dropout_grad_contract(L.ptr(slabs), 3, C.c_size_t(out * inn), L.ptr(U), L.ptr(Vs), Rp, out, inn, C.c_float(p), 11, 5,
                                           L.ptr(dU), L.ptr(dVs), L.ptr(scratch), L.stream()), "contract")
    keep = torch.from_numpy(keep_mask(out, inn, p, seed=11, linear_id=5)).to(DEV)
    g = dW.double() * keep.double() / (1.0 - p)
    for got, ref, what in ((dVs, g @ U.double(), "dVs"), (dU, g.t() @ Vs.double(), "dU")):
        err = (got.double() - ref).abs()
        assert (err <= 1e-4 * ref.abs() + 1e-3).all(), (what, float(err.max()))
    dU2, dVs2 = torch.empty_like(dU), torch.empty_like(dVs)
    L.check(lib.cara_dropout_grad_contract(L.ptr(slabs), 3, C.c_size_t(out * inn), L.ptr(U), L.ptr(Vs), Rp, out, inn, C.c_float(p), 11, 5,
                                           L.ptr(dU2), L.ptr(dVs2), L.ptr(scratch), L.stream()), "contract")
    assert torch.equal(dU, dU2) and torch.equal(dVs, dVs2)          # fixed summation order
    for M in (333, 5000, 100):
        X = _rnd(M, out + 8, seed=4)
        cs = torch.empty(out, device=DEV)
        csc = torch.empty(lib.cara_colsum_scratch_bytes(out), dtype=torch.uint8, device=DEV)
        L.check(lib.cara_colsum_bf16(L.ptr(X), out + 8, M, out, L.ptr(cs), L.ptr(csc), L.stream()), "colsum")
        assert torch.allclose(cs.double(), X[:, :out].double().sum(0), rtol=1e-5, atol=2e-3)
