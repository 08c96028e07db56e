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
    scratch = torch.empty(lib.cara_dropout_grad_scratch_bytes(inn, Rp), dtype=torch.uint8, device=DEV)
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
