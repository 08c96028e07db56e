"""GPU parity of the 160 x 256 x 64 one-workgroup-per-CU GEMM tile (cara_amd/csrc/gemm8.hip) against the 128 x 128 x 32
kernel it replaces on the long-K, narrow-N products (cara.py:87 fc2 forward, the dX of :75 fc1 and :25 qkv) -- bit for bit:
both kernels add the same 32-deep MFMA steps in the same order into fp32 accumulators and share the epilogues -- and against
fp64 on the same seeded inputs.  Through the C ABI (cara_gemm_bf16 / cara_gemm_with_tskinny_r); the tile family is picked per
call with the library's measurement switch."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def L():
    from cara_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, ref, rtol, atol, what=""):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.3e}"


@pytest.fixture(autouse=True)
def _policy_restored():
    yield
    L().lib().cara_debug_set_gemm8(-1)
    L().lib().cara_debug_set_gemm8_helpers(-1)


def _run(which, M, N, K, epi, mode, rank=16, seed=0, helpers=1):
    """One product on tile family `which` (0: the 128 x 128 x 32 kernel, 160: gemm8).  mode: 0 plain, 1 T given, 2 adapter inside
    (helpers = 1: T by the workgroup's helper waves, 0: by the tile waves)."""
    lib = L().lib()
    lib.cara_debug_set_gemm8(which)
    lib.cara_debug_set_gemm8_helpers(helpers)
    Rp = 32
    A, W = rnd(M, K, seed=seed + 1), rnd(N, K, seed=seed + 2, scale=0.05)
    Ut, Vs = rnd(Rp, K, seed=seed + 3, scale=0.1), rnd(N, Rp, seed=seed + 4, scale=0.3)
    Ut[rank:] = 0
    Vs[:, rank:] = 0
    T = rnd(M, Rp, seed=seed + 5)
    T[:, rank:] = 0
    bias = rnd(N, seed=seed + 6, dtype=torch.float32)
    kw = {}
    ldt = (M + 31) // 32 * 32
    To = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Tto = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
    if mode == 1:
        kw.update(A2=T, B2=Vs)
    elif mode == 2:
        kw.update(B2=Vs, Ut=Ut, T_out=To, Tt_out=Tto, Ut_rank=rank)
    extra = None
    if epi == "bf16":
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().gemm(A, W, out, epi=L().EPI_BF16, bias=bias, **kw)
    elif epi == "f32":
        out = torch.full((M, N), float("nan"), device=DEV)
        L().gemm(A, W, out, epi=L().EPI_F32, bias=bias, **kw)
    elif epi == "gelu":
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        extra = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().gemm(A, W, out, epi=L().EPI_GELU, bias=bias, C2=extra, **kw)
    elif epi == "dgelu":
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().gemm(A, W, out, epi=L().EPI_DGELU, aux=rnd(M, N, seed=seed + 7), **kw)
    else:
        out = torch.full((M, N), float("nan"), device=DEV)
        rs = (torch.arange((M + 196) // 197, device=DEV) % 3).float() * 0.55   # (some samples dropped: scale 0)
        L().gemm(A, W, out, epi=L().EPI_RESID, bias=bias, aux=rnd(M, N, seed=seed + 7, dtype=torch.float32), rowscale=rs,
                 rows_per_sample=197, **kw)
    res = [out] + ([extra] if extra is not None else []) + ([To, Tto[:, :M].clone()] if mode == 2 else [])
    ref = None
    if epi in ("bf16", "f32"):
        ref = A.double() @ W.double().t() + bias.double()
        if mode == 1:
            ref = ref + T.double() @ Vs.double().t()
        elif mode == 2:   # (with the device's own bf16 T as the operand: T itself is checked against fp64 by the caller)
            ref = ref + To.double() @ Vs.double().t()
            close(To, A.double() @ Ut.double().t(), 2 ** -8, 1e-3 * math.sqrt(K / 64), "T inside")
    return res, ref


# M = 12608: the headline rows (79 tiles of 160, the last one 128 rows); 4112 = 25 tiles + a 112-row edge tile whose second wave
# row holds 32 valid rows; N = 768 (three column tiles), 1024; every K of the model's long-K products, and K = 128 (two K steps:
# prologue and tail only)
@pytest.mark.parametrize("M,N,K,epi,mode", [
    (12608, 768, 3072, "resid", 2), (12608, 768, 3072, "bf16", 2), (12608, 768, 2304, "bf16", 2), (12608, 768, 3072, "bf16", 1),
    (4112, 768, 3072, "resid", 1), (4112, 1024, 2048, "bf16", 0), (4112, 768, 768, "gelu", 1), (4112, 768, 768, "dgelu", 2),
    (4112, 768, 128, "f32", 2), (4112, 768, 192, "bf16", 1), (4096, 256, 4096, "f32", 0), (4112, 784, 2048, "bf16", 1),
])
@pytest.mark.parametrize("helpers", [1, 0])
def test_gemm8_is_bitwise_the_128_tile_kernel(M, N, K, epi, mode, helpers):
    if helpers == 0 and mode != 2:
        pytest.skip("helper waves only differ with the adapter inside")
    old, ref = _run(0, M, N, K, epi, mode)
    new, _ = _run(160, M, N, K, epi, mode, helpers=helpers)
    for i, (x, y) in enumerate(zip(old, new)):
        assert not torch.isnan(y.float()).any(), f"output {i}: unwritten elements"
        assert torch.equal(x, y), f"output {i} differs: max |d| = {(x.float() - y.float()).abs().max().item():.3e}"
    if ref is not None:
        close(new[0], ref, 2 ** -8 if epi == "bf16" else 1e-5, 2e-3 * math.sqrt(K / 64), "against fp64")
    if mode == 2:   # T (bf16) against fp64, its transpose, and the zero columns beyond the rank
        assert torch.equal(new[-1], new[-2].t()) and torch.count_nonzero(new[-2][:, 16:]) == 0


def test_gemm8_rank_below_16_and_what_it_refuses():
    old, _ = _run(0, 4112, 768, 2048, "bf16", 2, rank=5)
    new, _ = _run(160, 4112, 768, 2048, "bf16", 2, rank=5)
    assert all(torch.equal(x, y) for x, y in zip(old, new))
    # what the tile does not take falls back to the 128 x 128 x 32 kernel (same results either way): a row count that is not a
    # multiple of 16, K-panel-major A, a second B operand
    for M in (4100,):
        old, _ = _run(0, M, 768, 2048, "bf16", 1)
        new, _ = _run(160, M, 768, 2048, "bf16", 1)
        assert all(torch.equal(x, y) for x, y in zip(old, new))


@pytest.mark.parametrize("M,N,K,epi,inside,Mts,first", [(12608, 768, 3072, "bf16", True, 12608, True), (12608, 768, 2304, "bf16", True, 12608, True),
                                                        (4112, 768, 2048, "bf16", False, 700, False), (4112, 768, 2048, "dgelu", False, 4112, True)])
def test_gemm8_carrying_transposed_skinny_products(M, N, K, epi, inside, Mts, first):
    """cara_gemm_with_tskinny_r on the tile: the dX product (with its own G' = dY Vs inside or given) and a pair of transposed
    skinny products of one r-tile (rank <= 16) as ONE grid -- the products' blocks as 512-thread workgroups behind the tiles.
    Bitwise what the 128 x 128 x 32 kernel's launch gives, and the products against fp64."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    Rp, rank = 32, 16
    dY, Wt = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Vst, U = rnd(Rp, K, seed=3, scale=0.1), rnd(N, Rp, seed=5, scale=0.3)
    Vst[rank:] = 0
    U[:, rank:] = 0
    Gin = rnd(M, Rp, seed=4)
    Gin[:, rank:] = 0
    K1a, K1b = 3072, 768
    Xo, dYo = rnd(Mts, K1a, seed=6), rnd(Mts, K1b, seed=7)
    ldg = (Mts + 31) // 32 * 32
    Gt, Tt = (torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    Gt[:rank, :Mts] = rnd(Mts, rank, seed=8, scale=0.5).t()
    Tt[:rank, :Mts] = rnd(Mts, rank, seed=9, scale=0.5).t()
    ldt = (M + 31) // 32 * 32
    aux = rnd(M, N, seed=10)

    def run(which, helpers=0):
        lib.cara_debug_set_gemm8(which)
        lib.cara_debug_set_gemm8_helpers(helpers)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        G = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
        Gto = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
        sa = torch.full((int(lib.cara_tskinny_scratch_bytes(Mts, K1a, Rp)) // 4,), float("nan"), device=DEV)
        sb = torch.full((int(lib.cara_tskinny_scratch_bytes(Mts, K1b, Rp)) // 4,), float("nan"), device=DEV)
        a = L().GemmArgs()
        a.A, a.lda, a.B, a.ldb, a.B2, a.Rp = p(dY), K, p(Wt), K, p(U), Rp
        if inside:
            a.Ut, a.T_out, a.Tt_out, a.ldt, a.Ut_rank = p(Vst), p(G), p(Gto), ldt, rank
        else:
            a.A2 = p(Gin)
        a.M, a.N, a.K, a.C, a.ldc = M, N, K, p(out), N
        a.epi = L().EPI_BF16 if epi == "bf16" else L().EPI_DGELU
        if epi == "dgelu":
            a.aux = p(aux)
        fmt = int(lib.cara_gemm_rider_slab_format(C.byref(a), Rp, rank))
        L().check(lib.cara_gemm_with_tskinny_r(C.byref(a), p(Xo) if first else None, K1a, p(Gt) if first else None, p(sa) if first else None, K1a,
                                               p(dYo), K1b, p(Tt), p(sb), K1b, 1, ldg, Mts, Rp, rank, st()), "cara_gemm_with_tskinny_r")
        # the products, reduced as the format says
        Db, cs = torch.empty(K1b, Rp, device=DEV), torch.empty(K1b, device=DEV)
        red = (L().TsReduce * 1)(L().TsReduce(p(sb), 0, p(Db), p(cs), 1, Mts, K1b, Rp, 16, fmt))
        L().check(lib.cara_tskinny_reduce_many(red, 1, st()), "reduce")
        Da = None
        if first:
            Da = torch.empty(K1a, Rp, device=DEV)
            red = (L().TsReduce * 1)(L().TsReduce(p(sa), 0, p(Da), None, 1, Mts, K1a, Rp, 16, fmt))
            L().check(lib.cara_tskinny_reduce_many(red, 1, st()), "reduce")
        return fmt, [out] + ([G, Gto[:, :M].clone()] if inside else []), [sa, sb], [Db, cs] + ([Da] if first else [])

    f0, old, old_slabs, old_red = run(0)
    f1, mid, mid_slabs, mid_red = run(160, helpers=0)
    f2, new, _, new_red = run(160, helpers=1)
    assert (f0, f1, f2) == (0, 0, 1)
    # the product itself (and G' inside): bit for bit on both forms; the riders: bit for bit where they run as workgroups
    # behind the tiles (same blocks, same order), to fp32 summation order where helper waves stream them (a slab per wave)
    for i, (x, y, z) in enumerate(zip(old, mid, new)):
        assert torch.equal(x, y) and torch.equal(x, z), f"output {i} differs"
    used = lambda t: torch.nan_to_num(t, nan=0.0)
    assert all(torch.equal(used(x), used(y)) for x, y in zip(old_slabs, mid_slabs))
    assert all(torch.equal(x, y) for x, y in zip(old_red, mid_red))
    for x, z in zip(old_red, new_red):
        assert not torch.isnan(z).any()
        close(z, x, 1e-5, 1e-4 * math.sqrt(Mts / 1000), "helper-streamed products against the block form")
    Db, cs = new_red[0], new_red[1]
    close(Db[:, :rank], dYo.double().t() @ Tt[:rank, :Mts].double().t(), 1e-3, 2e-2 * math.sqrt(Mts / 1000), "carried dVs")
    assert torch.count_nonzero(Db[:, rank:]) == 0
    close(cs, dYo.double().sum(0), 1e-3, 1e-2 * math.sqrt(Mts / 1000), "carried column sums")
    if first:
        close(new_red[2][:, :rank], Xo.double().t() @ Gt[:rank, :Mts].double().t(), 1e-3, 2e-2 * math.sqrt(Mts / 1000), "carried dU")


@pytest.mark.parametrize("M,N,K,inside", [(12608, 768, 3072, True), (12608, 768, 2304, True), (2000, 768, 768, False), (12608, 768, 3072, False)])
def test_dvs_out_of_the_dx_tiles_own_a_sub_buffers(M, N, K, inside):
    """cara_gemm_args::er_Tt with CARA_EPI_BF16 (cara_gemm_dv_chunks): the dX launch on the 160 x 256 x 64 tile leaves dVs = A^T T and the
    column sums of A from the A tiles of its own K loop.  C (and T with the adapter inside) bitwise the launch without it; the reduced
    products against fp64 and against cara_tskinny_xtg on the same operands.  M = 2000: the last row tile holds 80 rows."""
    Lm = L()
    lib = Lm.lib()
    p, st = Lm.ptr, Lm.stream
    Rp, rank = 32, 16
    ldg = (M + 31) // 32 * 32
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    B2 = rnd(N, Rp, seed=4, scale=0.3)
    Tt = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Tt[:rank, :M] = rnd(rank, M, seed=6)
    Tt[:, M:] = float("nan")
    kw = {}
    if inside:
        Ut = torch.zeros(Rp, K, dtype=torch.bfloat16, device=DEV)
        Ut[:rank] = rnd(rank, K, seed=3, scale=0.1)
        kw = dict(Ut=Ut, B2=B2, Ut_rank=rank)
        T0, Tt0 = torch.zeros(M, Rp, dtype=torch.bfloat16, device=DEV), torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
        T1, Tt1 = torch.zeros_like(T0), torch.zeros_like(Tt0)
    else:
        kw = dict(A2=rnd(M, Rp, seed=3, scale=0.3), B2=B2)
    ref = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    lib.cara_debug_set_gemm8(160)
    try:
        Lm.gemm(A, W, ref, epi=Lm.EPI_BF16, **(dict(kw, T_out=T0, Tt_out=Tt0) if inside else kw))
        out = torch.zeros_like(ref)
        out, slabs, chunks = Lm.gemm(A, W, out, epi=Lm.EPI_BF16, dv=(Tt, True), **(dict(kw, T_out=T1, Tt_out=Tt1) if inside else kw))
    finally:
        lib.cara_debug_set_gemm8(-1)
    assert chunks == (M + 159) // 160
    assert torch.equal(out, ref)
    if inside:
        assert torch.equal(T0, T1) and torch.equal(Tt0[:, :M], Tt1[:, :M])
    D = torch.full((K, Rp), float("nan"), device=DEV)
    cs = torch.full((K,), float("nan"), device=DEV)
    tab = (Lm.TsReduce * 1)(Lm.TsReduce(p(slabs), 0, p(D), p(cs), 1, M, K, Rp, 16, chunks))
    Lm.check(lib.cara_tskinny_reduce_many(tab, 1, st()), "reduce many")
    torch.cuda.synchronize()
    assert torch.count_nonzero(D[:, rank:]) == 0
    close(D[:, :rank], A.double().t() @ Tt[:rank, :M].double().t(), 1e-3, 2e-2, "dVs from the A tiles")
    close(cs, A.double().sum(0), 1e-3, 2e-2, "column sums from the A tiles")
    D2, cs2 = torch.empty(K, Rp, device=DEV), torch.empty(K, device=DEV)
    Tz = Tt.clone()
    Tz[:, M:] = 0
    Lm.tskinny_xtg(A, Tz, D2, cs2, M=M)
    close(D[:, :rank], D2[:, :rank], 1e-5, 2e-3, "dVs: A tiles vs cara_tskinny_xtg")
    close(cs, cs2, 1e-5, 2e-3, "column sums: A tiles vs cara_tskinny_xtg")
