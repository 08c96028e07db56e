"""GPU parity tests of every C-ABI kernel against fp32/fp64 torch restatements of the same op on
the same seeded inputs.  Tolerances are stated per test: inputs are bf16-representable, products
are accumulated in fp32 on the device and in fp64 here, so differences come only from the output
rounding (bf16: 2^-9 relative) and fp32 summation order."""
import ctypes as C
import math
import os

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
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.3e} (ref max {ref.abs().max():.3e})"


# ------------------------------------------------------------------------------------------
# MFMA layout check with exact integer data (asymmetric operands): catches swapped maps
# ------------------------------------------------------------------------------------------
def test_gemm_exact_small_integers():
    _ = L()
    M, N, K = 128, 128, 64
    A = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
    B = (torch.arange(N * K).reshape(N, K) % 5 - 2).float() + (torch.arange(N).reshape(N, 1) % 3).float()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L().gemm(A.bfloat16().to(DEV), B.bfloat16().to(DEV), out, epi=L().EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), A @ B.t())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (12608, 768, 768), (333, 100, 768), (64, 2304, 128), (197, 3072, 192)])
def test_gemm_shapes_f32(M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias = rnd(N, seed=3, dtype=torch.float32)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    L().gemm(A, B, out, epi=L().EPI_F32, bias=bias)
    ref = A.double() @ B.double().t() + bias.double()
    close(out, ref, 1e-4, 1e-3 * math.sqrt(K / 64), f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("Rp", [32, 64])
def test_gemm_k_extension(Rp):
    M, N, K = 777, 640, 768
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    A2, B2 = rnd(M, Rp, seed=3), rnd(N, Rp, seed=4)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L().gemm(A, B, out, epi=L().EPI_F32, A2=A2, B2=B2)
    ref = A.double() @ B.double().t() + A2.double() @ B2.double().t()
    close(out, ref, 1e-4, 5e-3, "gemm+ext")


def test_gemm_epilogues():
    M, N, K = 400, 256, 128
    A, B = rnd(M, K, seed=1, scale=0.3), rnd(N, K, seed=2, scale=0.3)
    bias = rnd(N, seed=3, dtype=torch.float32)
    acc = A.double() @ B.double().t() + bias.double()
    # bf16 out
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L().gemm(A, B, out, epi=L().EPI_BF16, bias=bias)
    close(out, acc, 2 ** -8, 1e-3, "epi bf16")
    # gelu: u (bf16) and gelu(u)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L().gemm(A, B, h, epi=L().EPI_GELU, bias=bias, C2=u)
    close(u, acc, 2 ** -8, 1e-3, "epi gelu u")
    close(h, torch.nn.functional.gelu(acc), 2 ** -8, 1e-3, "epi gelu h")   # GELU of the unrounded accumulator
    # residual with per-sample scale: rows_per_sample = 50 -> 8 samples
    xin = rnd(M, N, seed=5, dtype=torch.float32)
    rs = torch.tensor([1.0, 0.0, 1.1, 1.1, 0.0, 1.1, 1.0, 1.1], device=DEV)
    xo = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L().gemm(A, B, xo, epi=L().EPI_RESID, bias=bias, aux=xin, rowscale=rs, rows_per_sample=50)
    ref = xin.double() + rs.double().repeat_interleave(50)[:, None] * acc
    close(xo, ref, 1e-5, 1e-4, "epi resid")
    # dgelu
    dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L().gemm(A, B, dg, epi=L().EPI_DGELU, aux=u)
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    close(dg, (A.double() @ B.double().t()) * ud.grad, 2 ** -8, 2e-3, "epi dgelu")


@pytest.mark.parametrize("M,N,K,inside", [(1300, 768, 256, False), (1300, 768, 256, True), (1700, 3072, 128, False)])
def test_gemm_residual_epilogue_of_interior_tiles(M, N, K, inside):
    """The residual epilogue of interior wave tiles (input rows requested a pass group ahead, the per-sample scale without a
    division per row): samples of 197 rows, so that wave tiles straddle sample boundaries; with and without T = X U inside."""
    rps, Rp, rank = 197, 32, 16
    A, B = rnd(M, K, seed=1, scale=0.3), rnd(N, K, seed=2, scale=0.3)
    bias = rnd(N, seed=3, dtype=torch.float32)
    xin = rnd(M, N, seed=5, dtype=torch.float32)
    rs = torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(7)).to(DEV) + 0.25
    Vs = rnd(N, Rp, seed=9, scale=0.3)
    Ut = torch.zeros(Rp, K, dtype=torch.bfloat16, device=DEV)
    Ut[:rank] = rnd(rank, K, seed=8, scale=0.3)
    T = (A.double() @ Ut.double().t()).bfloat16()
    acc = A.double() @ B.double().t() + bias.double() + T.double() @ Vs.double().t()
    ref = xin.double() + rs.double().repeat_interleave(rps)[:M, None] * acc
    xo = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    if inside:
        Tb = torch.empty(M, Rp, dtype=torch.bfloat16, device=DEV)
        L().gemm(A, B, xo, epi=L().EPI_RESID, bias=bias, aux=xin, rowscale=rs, rows_per_sample=rps, B2=Vs, Ut=Ut, T_out=Tb)
    else:
        L().gemm(A, B, xo, epi=L().EPI_RESID, bias=bias, aux=xin, rowscale=rs, rows_per_sample=rps, A2=T, B2=Vs)
    close(xo, ref, 1e-5, 2e-3, "resid interior")


@pytest.mark.parametrize("M,N", [(300, 200), (192, 328), (1200, 3080)])
def test_gemm_edge_tiles_mixing_epilogue_paths(M, N):
    """Edge tiles in which some waves own a complete 64-column (or row) sub-tile and take the fast bf16 epilogue while others
    take the generic one: the two paths stage through LDS and must not share a region (they once did)."""
    K = 256
    A, B = rnd(M, K, seed=1, scale=0.3), rnd(N, K, seed=2, scale=0.3)
    bias = rnd(N, seed=3, dtype=torch.float32)
    acc = A.double() @ B.double().t() + bias.double()
    ldc = (N + 7) // 8 * 8
    for _ in range(3):   # a race shows up intermittently
        out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=DEV)
        L().gemm(A, B, out, epi=L().EPI_BF16, bias=bias, ldc=ldc, N=N)
        close(out[:, :N], acc, 2 ** -8, 1e-3, "edge bf16")
        h, u = torch.zeros_like(out), torch.zeros_like(out)
        L().gemm(A, B, h, epi=L().EPI_GELU, bias=bias, C2=u, ldc=ldc, N=N)
        close(u[:, :N], acc, 2 ** -8, 1e-3, "edge gelu u")
        close(h[:, :N], torch.nn.functional.gelu(acc), 2 ** -8, 1e-3, "edge gelu h")


@pytest.mark.parametrize("M,N,K,rank", [(12608, 768, 3072, 16), (12608, 768, 768, 8), (1500, 3072, 768, 32), (333, 300, 128, 16),
                                        (12608, 768, 3072, 64), (12608, 768, 768, 48), (333, 300, 128, 33)])
def test_gemm_with_adapter_inside(M, N, K, rank):
    """cara_gemm_args.Ut: T = A Ut^T computed per tile inside the GEMM and used as the K-extension operand; must
    agree with cara_skinny_xu + the ordinary K-extension, and leave T / Tt behind for the backward.  Rank > 32: Rp = 64
    (two extension steps, eight T tiles per wave)."""
    Rp = 32 if rank <= 32 else 64
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Ut = rnd(Rp, K, seed=3, scale=0.1)
    Ut[rank:] = 0
    Vs = rnd(N, Rp, seed=4, scale=0.3)
    bias = rnd(N, seed=5, dtype=torch.float32)
    ldt = (M + 31) // 32 * 32
    T = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Tt = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    L().gemm(A, B, out, epi=L().EPI_F32, bias=bias, B2=Vs, Ut=Ut, T_out=T, Tt_out=Tt)
    Tref = A.double() @ Ut.double().t()
    close(T, Tref, 2 ** -8, 1e-3 * math.sqrt(K / 64), "T inside the GEMM")
    assert torch.equal(Tt[:, :M], T.t()) and torch.count_nonzero(Tt[:, M:]) == 0 and torch.count_nonzero(T[:, rank:]) == 0
    ref = A.double() @ B.double().t() + bias.double() + T.double() @ Vs.double().t()      # with the bf16 T it produced
    close(out, ref, 1e-4, 1e-3 * math.sqrt(K / 64) + 2e-3, "gemm with the adapter inside")
    T2 = torch.empty_like(T)
    L().skinny_xu(A, Ut, T2)
    out2 = torch.empty_like(out)
    L().gemm(A, B, out2, epi=L().EPI_F32, bias=bias, A2=T2, B2=Vs)
    # (the two T differ by one bf16 ulp where an fp32 sum sits on a rounding boundary: ~0.06 on values of ~10, times Vs)
    close(out, out2.double(), 1e-3, 0.2, "vs skinny + K-extension")
    # GELU epilogue through the same kernel
    h, u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L().gemm(A, B, h, epi=L().EPI_GELU, bias=bias, B2=Vs, C2=u, Ut=Ut, T_out=T, Tt_out=Tt)
    close(u, ref, 2 ** -8, 2e-3 * math.sqrt(K / 64) + 2e-3, "adapter inside, gelu u")
    close(h, torch.nn.functional.gelu(ref), 2 ** -8, 2e-3 * math.sqrt(K / 64) + 2e-3, "adapter inside, gelu h")


@pytest.mark.parametrize("M,N,K,Rp", [(12608, 768, 3072, 32), (1500, 3072, 768, 32), (333, 300, 128, 0), (4096, 2304, 768, 64)])
def test_gemm_b_in_k_panel_major_layout(M, N, K, Rp):
    """cara_gemm_args.Bp: the same B as [K/32][N][32] panels (cara_pack_b_panels).  Only the addresses the operand
    tiles are fetched from change, so every result is BITWISE what the row-major image gives -- plain, with the
    K-extension, with the adapter inside, ragged N."""
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Bp = L().pack_b_panels(B)
    assert torch.equal(Bp, B.view(N, K // 32, 32).permute(1, 0, 2).contiguous())
    bias = rnd(N, seed=5, dtype=torch.float32)
    kw = dict(epi=L().EPI_F32, bias=bias)
    if Rp:
        kw.update(A2=rnd(M, Rp, seed=3), B2=rnd(N, Rp, seed=4, scale=0.3))
    o1 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    o2 = torch.full_like(o1, float("nan"))
    L().gemm(A, B, o1, **kw)
    L().gemm(A, B, o2, Bp=Bp, **kw)
    assert torch.equal(o1, o2)
    ref = A.double() @ B.double().t() + bias.double()
    if Rp:
        ref = ref + kw["A2"].double() @ kw["B2"].double().t()
    close(o2, ref, 1e-4, 1e-3 * math.sqrt(K / 64) + 2e-3, "gemm with packed B")
    if Rp == 32:   # adapter inside
        Ut = rnd(32, K, seed=6, scale=0.1)
        T1, T2 = (torch.empty(M, 32, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        h1, h2, u1, u2 = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(4))
        L().gemm(A, B, h1, epi=L().EPI_GELU, bias=bias, B2=kw["B2"], C2=u1, Ut=Ut, T_out=T1)
        L().gemm(A, B, h2, epi=L().EPI_GELU, bias=bias, B2=kw["B2"], C2=u2, Ut=Ut, T_out=T2, Bp=Bp)
        assert torch.equal(h1, h2) and torch.equal(u1, u2) and torch.equal(T1, T2)


def _panels(X):
    """[M, K] -> K-panel-major [K/32, M, 32]"""
    M, K = X.shape
    return X.view(M, K // 32, 32).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("M,N,K", [(12608, 3072, 768), (12608, 768, 3072), (1000, 768, 2304)])
def test_gemm_activations_in_k_panel_major_layout(M, N, K):
    """cara_gemm_args.a_panels / c_panels: A read from, C written as, [K/32][M][32] panels.  Addresses only: every
    output is BITWISE the row-major result (re-laid), for the plain, GELU, GELU' and adapter-inside forms."""
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Ap, Bp = _panels(A), L().pack_b_panels(B)
    A2, B2 = rnd(M, 32, seed=3), rnd(N, 32, seed=4, scale=0.3)
    bias = rnd(N, seed=5, dtype=torch.float32)
    o1, o2 = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    op = torch.full((N // 32, M, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    L().gemm(A, B, o1, epi=L().EPI_BF16, bias=bias, A2=A2, B2=B2, Bp=Bp)
    L().gemm(Ap, B, o2, epi=L().EPI_BF16, bias=bias, A2=A2, B2=B2, Bp=Bp, a_panels=M, M=M, K=K, lda=K)
    assert torch.equal(o1, o2)
    L().gemm(Ap, B, op, epi=L().EPI_BF16, bias=bias, A2=A2, B2=B2, Bp=Bp, a_panels=M, c_panels=M, M=M, K=K, lda=K, ldc=N)
    assert torch.equal(op, _panels(o1))
    # GELU: h as panels, u row-major; GELU': aux row-major, output as panels
    h1, u1, u2 = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(3))
    L().gemm(A, B, h1, epi=L().EPI_GELU, bias=bias, A2=A2, B2=B2, C2=u1)
    L().gemm(A, B, op, epi=L().EPI_GELU, bias=bias, A2=A2, B2=B2, C2=u2, c_panels=M, ldc=N)
    assert torch.equal(op, _panels(h1)) and torch.equal(u1, u2)
    L().gemm(A, B, o1, epi=L().EPI_DGELU, A2=A2, B2=B2, aux=u1)
    L().gemm(Ap, B, op, epi=L().EPI_DGELU, A2=A2, B2=B2, aux=u1, a_panels=M, c_panels=M, M=M, K=K, lda=K, ldc=N)
    assert torch.equal(op, _panels(o1))
    # adapter inside the GEMM on a panel-major A
    Ut = rnd(32, K, seed=6, scale=0.1)
    T1, T2 = (torch.empty(M, 32, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    f1, f2 = (torch.empty(M, N, dtype=torch.float32, device=DEV) for _ in range(2))
    L().gemm(A, B, f1, epi=L().EPI_F32, bias=bias, B2=B2, Ut=Ut, T_out=T1)
    L().gemm(Ap, B, f2, epi=L().EPI_F32, bias=bias, B2=B2, Ut=Ut, T_out=T2, Bp=Bp, a_panels=M, M=M, K=K, lda=K)
    assert torch.equal(f1, f2) and torch.equal(T1, T2)
    # refused where the layout is not implemented: few rows (the split-K path), fp32 output as panels
    with pytest.raises(L().CaraError):
        L().gemm(Ap[:, :64], B, o2[:64], epi=L().EPI_BF16, a_panels=64, M=64, K=K, lda=K)
    with pytest.raises(L().CaraError):
        L().gemm(A, B, f1, epi=L().EPI_F32, c_panels=M)


@pytest.mark.parametrize("M,K", [(12608, 3072), (12608, 768), (1000, 2304), (2000, 4096), (50, 1024)])
def test_skinny_products_on_k_panel_major_operands(M, K):
    """cara_skinny_xu / cara_tskinny_xtg with ldx = -M: the same products, bitwise, from the panel image of X."""
    X = rnd(M, K, seed=1)
    Xp = _panels(X)
    Ut = rnd(32, K, seed=2, scale=0.1)
    ldt = (M + 31) // 32 * 32
    T1, T2 = (torch.empty(M, 32, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    Tt1, Tt2 = (torch.zeros(32, ldt, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    L().skinny_xu(X, Ut, T1, Tt1)
    L().skinny_xu(Xp, Ut, T2, Tt2, panels=True)
    assert torch.equal(T1, T2) and torch.equal(Tt1, Tt2)
    D1, D2 = (torch.empty(K, 32, dtype=torch.float32, device=DEV) for _ in range(2))
    c1, c2 = (torch.empty(K, dtype=torch.float32, device=DEV) for _ in range(2))
    L().tskinny_xtg(X, Tt1, D1, c1)
    L().tskinny_xtg(Xp, Tt1, D2, c2, panels=True)
    assert torch.equal(D1, D2) and torch.equal(c1, c2)


def _sk_scratch():
    return torch.empty(L().gemm_scratch_bytes(), dtype=torch.uint8, device=DEV)


@pytest.mark.parametrize("M,N,K,Rp", [(64, 768, 3072, 32), (64, 3072, 768, 32), (64, 768, 768, 64), (17, 300, 2304, 0), (128, 768, 768, 32)])
def test_gemm_few_rows_split_k(M, N, K, Rp):
    """Few-row products with caller scratch: K slabs in one batched launch + a finishing kernel (bias, rank-R term,
    epilogue).  Rows strided as the last block's cls rows are (row stride 5 * width)."""
    sc = _sk_scratch()
    stride = 5
    Abig, B = rnd(M * stride, K, seed=1, scale=0.3), rnd(N, K, seed=2, scale=0.3)
    A = Abig.view(M, stride * K)[:, :K]                     # row stride 5K
    bias = rnd(N, seed=3, dtype=torch.float32)
    A2 = rnd(M, Rp, seed=4) if Rp else None
    B2 = rnd(N, Rp, seed=5, scale=0.3) if Rp else None
    acc = A.double() @ B.double().t() + bias.double()
    if Rp:
        acc = acc + A2.double() @ B2.double().t()
    kw = dict(bias=bias, A2=A2, B2=B2, scratch=sc, lda=stride * K, M=M, K=K)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    L().gemm(Abig, B, out, epi=L().EPI_F32, **kw)
    close(out, acc, 1e-4, 2e-3 * math.sqrt(K / 64), "few rows f32")
    h, u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L().gemm(Abig, B, h, epi=L().EPI_GELU, C2=u, **kw)
    close(u, acc, 2 ** -8, 2e-3 * math.sqrt(K / 64), "few rows gelu u")
    close(h, torch.nn.functional.gelu(acc), 2 ** -8, 2e-3 * math.sqrt(K / 64), "few rows gelu h")
    # residual epilogue into a strided fp32 stream (ldc = 3N), per-row scale
    xin = rnd(M, 3 * N, seed=6, dtype=torch.float32)
    xo = xin.clone()
    rs = (torch.arange(M, device=DEV) % 3).float() * 0.5 + 0.5
    L().gemm(Abig, B, xo, epi=L().EPI_RESID, aux=xin, rowscale=rs, rows_per_sample=1, ldc=3 * N, N=N, **kw)
    close(xo[:, :N], xin[:, :N].double() + rs.double()[:, None] * acc, 1e-5, 2e-3 * math.sqrt(K / 64), "few rows resid")
    assert torch.equal(xo[:, N:], xin[:, N:])
    dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    kw2 = dict(kw)
    kw2["bias"] = None
    L().gemm(Abig, B, dg, epi=L().EPI_DGELU, aux=u, **kw2)
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    close(dg, (acc - bias.double()) * ud.grad, 2 ** -8, 3e-3 * math.sqrt(K / 64), "few rows dgelu")


@pytest.mark.parametrize("Kr,M,N,nslab", [(12608, 768, 768, 4), (12608, 3072, 768, 4), (12608, 768, 2304, 4), (1000, 256, 128, 1),
                                          (333, 128, 384, 2), (64, 128, 128, 2)])
def test_gemm_tn_from_row_major_operands(Kr, M, N, nslab):
    """cara_gemm_tn_f32: C[z] = At[rows of slab z]^T Bt[rows of slab z] (dW = dY^T X) from ROW-major operands through
    transposing LDS reads; the slabs sum to the full product, each slab matches its own row range, ragged last steps."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    At, Bt = rnd(Kr, M + 8, seed=1), rnd(Kr, N + 16, seed=2)           # row strides larger than the widths
    out = torch.full((nslab, M, N), float("nan"), dtype=torch.float32, device=DEV)
    L().check(lib.cara_gemm_tn_f32(p(At), M + 8, p(Bt), N + 16, p(out), N, M, N, Kr, nslab, C.c_size_t(M * N), st()), "gemm_tn")
    kslab = ((Kr + nslab - 1) // nslab + 31) // 32 * 32
    for z in range(nslab):
        a, b = At[z * kslab:(z + 1) * kslab, :M].double(), Bt[z * kslab:(z + 1) * kslab, :N].double()
        close(out[z], a.t() @ b, 1e-5, 1e-3 * math.sqrt(kslab / 64), f"slab {z}")
    close(out.double().sum(0), At[:, :M].double().t() @ Bt[:, :N].double(), 1e-5, 2e-3 * math.sqrt(Kr / 64), "sum of slabs")
    # the same numbers as the transposed-copies route (cara_transpose_bf16_ld + batched cara_gemm_bf16) up to fp32 summation order
    assert lib.cara_gemm_tn_f32(p(At), M + 8, p(Bt), N + 16, p(out), N, M - 8, N, Kr, nslab, C.c_size_t(M * N), st()) != 0   # M % 128
    assert lib.cara_gemm_tn_f32(p(At), M + 8, p(Bt), N + 16, p(out), N, M, N, 32, 2, C.c_size_t(M * N), st()) != 0         # an empty slab


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,Rp", [(12608, 768, 32), (12608, 3072, 32), (197, 768, 64), (45, 2304, 32),
                                    (12608, 3072, 64), (12608, 2304, 64), (1001, 768, 64)])
def test_skinny_xu(M, K, Rp):
    X, Ut = rnd(M, K, seed=1), rnd(Rp, K, seed=2, scale=0.1)
    ldt = (M + 31) // 32 * 32
    T = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Tt = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
    L().skinny_xu(X, Ut, T, Tt)
    ref = X.double() @ Ut.double().t()
    close(T, ref, 2 ** -8, 1e-3, "skinny T")
    close(Tt[:, :M], ref.t(), 2 ** -8, 1e-3, "skinny Tt")
    assert torch.count_nonzero(Tt[:, M:]) == 0


@pytest.mark.parametrize("M,K1,Rp", [(12608, 768, 32), (12608, 3072, 32), (12608, 2304, 64), (197, 768, 32), (70, 128, 32)])
def test_tskinny(M, K1, Rp):
    X = rnd(M, K1, seed=1)
    ldg = (M + 31) // 32 * 32
    G = rnd(M, Rp, seed=2, scale=0.5)
    Gt = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Gt[:, :M] = G.t()
    D = torch.full((K1, Rp), float("nan"), dtype=torch.float32, device=DEV)
    cs = torch.full((K1,), float("nan"), dtype=torch.float32, device=DEV)
    L().tskinny_xtg(X, Gt, D, cs)
    close(D, X.double().t() @ G.double(), 1e-4, 2e-3 * math.sqrt(M / 100), "tskinny D")
    close(cs, X.double().sum(0), 1e-4, 2e-3 * math.sqrt(M / 100), "tskinny colsum")
    D2 = torch.empty_like(D)
    L().tskinny_xtg(X, Gt, D2, None)
    assert torch.equal(D, D2), "tskinny must be bitwise reproducible (fixed-order slab sum)"


@pytest.mark.parametrize("M,N,K,epi,Rp", [(12608, 768, 3072, "bf16", 32), (12608, 3072, 768, "dgelu", 32), (1500, 768, 768, "bf16", 32),
                                          (12608, 768, 3072, "bf16", 64), (12608, 3072, 768, "dgelu", 64), (1500, 2304, 768, "bf16", 64)])
def test_gemm_carrying_the_transposed_skinny_products(M, N, K, epi, Rp, monkeypatch):
    """cara_gemm_with_tskinny: the dX GEMM of a linear and its two transposed skinny products as ONE launch give
    bitwise what cara_gemm_bf16 + cara_tskinny_partial2 give (Rp = 64: the products' 16 accumulator tiles, combined
    in two passes)."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    dY, Wt = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)          # dX = dY Wt^T: K = out features, N = in features
    X = rnd(M, N, seed=3)
    G, U = rnd(M, Rp, seed=4, scale=0.5), rnd(N, Rp, seed=5, scale=0.3)
    ldg = (M + 31) // 32 * 32
    Gt, Tt = (torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    Gt[:, :M] = G.t()
    Tt[:, :M] = rnd(M, Rp, seed=6, scale=0.5).t()
    aux = rnd(M, N, seed=7)

    def run(fused):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, N, Rp)), dtype=torch.uint8, device=DEV)
        sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, K, Rp)), dtype=torch.uint8, device=DEV)
        a = L().GemmArgs()
        a.A, a.lda, a.B, a.ldb, a.A2, a.B2, a.Rp = p(dY), K, p(Wt), K, p(G), p(U), Rp
        a.M, a.N, a.K, a.C, a.ldc = M, N, K, p(out), N
        a.epi = L().EPI_DGELU if epi == "dgelu" else L().EPI_BF16
        a.aux = p(aux) if epi == "dgelu" else None
        ts = (p(X), N, p(Gt), p(sa), N, p(dY), K, p(Tt), p(sb), K, 1, ldg, M, Rp, st())
        if fused:
            L().check(lib.cara_gemm_with_tskinny(C.byref(a), *ts), "cara_gemm_with_tskinny")
        else:
            L().check(lib.cara_gemm_bf16(C.byref(a), st()), "gemm")
            L().check(lib.cara_tskinny_partial2(*ts), "partial2")
        return out, sa, sb

    ref = run(False)
    got = run(True)
    assert all(torch.equal(x, y) for x, y in zip(ref, got))
    # ... and right: the products' slabs, summed, against fp64 (dU = X^T G, dVs = dY^T T, dc = colsum dY)
    D = torch.empty(N, Rp, device=DEV)
    L().check(lib.cara_tskinny_reduce(p(got[1]), C.c_size_t(0), p(D), None, 1, M, N, Rp, st()), "reduce dU")
    close(D, X.double().t() @ G.double(), 1e-3, 2e-2 * math.sqrt(M / 1000), "riding dU")
    D2, cs = torch.empty(K, Rp, device=DEV), torch.empty(K, device=DEV)
    L().check(lib.cara_tskinny_reduce(p(got[2]), C.c_size_t(0), p(D2), p(cs), 1, M, K, Rp, st()), "reduce dVs")
    close(D2, dY.double().t() @ Tt[:, :M].double().t(), 1e-3, 2e-2 * math.sqrt(M / 1000), "riding dVs")
    close(cs, dY.double().sum(0), 1e-3, 2e-2 * math.sqrt(M / 1000), "riding dc")
    # the same with dY and X K-panel-major (a_panels on the GEMM, negative ld on the products) and packed weights
    dYp, Xp, Wp = _panels(dY), _panels(X), L().pack_b_panels(Wt)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, N, Rp)), dtype=torch.uint8, device=DEV)
    sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, K, Rp)), dtype=torch.uint8, device=DEV)
    a = L().GemmArgs()
    a.A, a.lda, a.a_panels, a.B, a.ldb, a.Bp, a.A2, a.B2, a.Rp = p(dYp), 0, M, p(Wt), K, p(Wp), p(G), p(U), Rp
    a.M, a.N, a.K, a.C, a.ldc = M, N, K, p(out), N
    a.epi = L().EPI_DGELU if epi == "dgelu" else L().EPI_BF16
    a.aux = p(aux) if epi == "dgelu" else None
    L().check(lib.cara_gemm_with_tskinny(C.byref(a), p(Xp), -M, p(Gt), p(sa), N, p(dYp), -M, p(Tt), p(sb), K, 1, ldg, M, Rp, st()),
              "cara_gemm_with_tskinny, panels")
    assert all(torch.equal(x, y) for x, y in zip(ref, (out, sa, sb))), "panel-major operands"
    # not fusable: few rows
    a = L().GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.M, a.N, a.K, a.ldc = p(dY), K, p(Wt), K, 64, N, K, N
    out = torch.empty(64, N, dtype=torch.bfloat16, device=DEV)
    a.C, a.epi = p(out), L().EPI_BF16
    sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(64, N, Rp)), dtype=torch.uint8, device=DEV)
    sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(64, K, Rp)), dtype=torch.uint8, device=DEV)
    assert lib.cara_gemm_with_tskinny(C.byref(a), p(X), N, p(Gt), p(sa), N, p(dY), K, p(Tt), p(sb), K, 0, ldg, 64, Rp, st()) != 0


@pytest.mark.parametrize("M,N,K,Rp,Mts", [(12608, 768, 3072, 32, 12608), (12608, 768, 2304, 64, 12608), (1500, 768, 768, 32, 700)])
def test_adapter_inside_gemm_carrying_another_linears_products(M, N, K, Rp, Mts):
    """cara_gemm_with_tskinny with cara_gemm_args.Ut: the dX GEMM that computes its own G' = dY Vs inside carries the
    transposed skinny products of ANOTHER linear (deferred products: their row count may differ); bitwise what the two
    separate launches give."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    dY, Wt = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Vst, U = rnd(Rp, K, seed=3, scale=0.1), rnd(N, Rp, seed=5, scale=0.3)
    K1a, K1b = 768, 2304                                   # the carried pair: another linear's X [Mts, K1a] and dY [Mts, K1b]
    Xo, dYo = rnd(Mts, K1a, seed=6), rnd(Mts, K1b, seed=7)
    ldg = (Mts + 31) // 32 * 32
    Gt, Tt = (torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    Gt[:, :Mts] = rnd(Mts, Rp, seed=8, scale=0.5).t()
    Tt[:, :Mts] = rnd(Mts, Rp, seed=9, scale=0.5).t()
    ldt = (M + 31) // 32 * 32

    def run(fused):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        G = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
        Gto = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
        sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(Mts, K1a, Rp)), dtype=torch.uint8, device=DEV)
        sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(Mts, K1b, Rp)), dtype=torch.uint8, device=DEV)
        a = L().GemmArgs()
        a.A, a.lda, a.B, a.ldb, a.B2, a.Rp = p(dY), K, p(Wt), K, p(U), Rp
        a.Ut, a.T_out, a.Tt_out, a.ldt = p(Vst), p(G), p(Gto), ldt
        a.M, a.N, a.K, a.C, a.ldc, a.epi = M, N, K, p(out), N, L().EPI_BF16
        ts = (p(Xo), K1a, p(Gt), p(sa), K1a, p(dYo), K1b, p(Tt), p(sb), K1b, 1, ldg, Mts, Rp, st())
        if fused:
            L().check(lib.cara_gemm_with_tskinny(C.byref(a), *ts), "cara_gemm_with_tskinny (adapter inside)")
        else:
            L().check(lib.cara_gemm_bf16(C.byref(a), st()), "gemm")
            L().check(lib.cara_tskinny_partial2(*ts), "partial2")
        return out, G, Gto, sa, sb

    ref, got = run(False), run(True)
    assert all(torch.equal(x, y) for x, y in zip(ref, got))
    close(got[1], dY.double() @ Vst.double().t(), 2 ** -8, 1e-3 * math.sqrt(K / 64), "G' inside")
    D = torch.empty(K1b, Rp, device=DEV)
    L().check(lib.cara_tskinny_reduce(p(got[4]), C.c_size_t(0), p(D), None, 1, Mts, K1b, Rp, st()), "reduce")
    close(D, dYo.double().t() @ Tt[:, :Mts].double().t(), 1e-3, 2e-2 * math.sqrt(Mts / 1000), "carried dVs")
    # a non-bf16 epilogue with the adapter inside cannot carry products
    a = L().GemmArgs()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    G = torch.empty(M, Rp, dtype=torch.bfloat16, device=DEV)
    a.A, a.lda, a.B, a.ldb, a.B2, a.Rp, a.Ut, a.T_out = p(dY), K, p(Wt), K, p(U), Rp, p(Vst), p(G)
    a.M, a.N, a.K, a.C, a.ldc, a.epi = M, N, K, p(out), N, L().EPI_F32
    sa = torch.zeros(int(lib.cara_tskinny_scratch_bytes(Mts, K1a, Rp)), dtype=torch.uint8, device=DEV)
    sb = torch.zeros(int(lib.cara_tskinny_scratch_bytes(Mts, K1b, Rp)), dtype=torch.uint8, device=DEV)
    assert lib.cara_gemm_with_tskinny(C.byref(a), p(Xo), K1a, p(Gt), p(sa), K1a, p(dYo), K1b, p(Tt), p(sb), K1b, 1, ldg, Mts, Rp, st()) != 0


def test_tskinny_reductions_in_one_launch():
    """cara_tskinny_reduce_many: several slab reductions of different shapes (and layer counts) in one launch give
    bitwise what one cara_tskinny_reduce launch per product gives."""
    lib = L().lib()
    p, st = L().ptr, L().stream

    Red = L().TsReduce       # (the mirror the library's struct size is checked against at load time; Rc stays 0 = all columns)

    lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    probs, keep = [], []
    for (M, K1, batch, want_cs, seed) in [(3000, 768, 3, False, 1), (3000, 3072, 3, True, 2), (64, 768, 1, True, 3), (500, 2304, 2, False, 4)]:
        nb = int(lib.cara_tskinny_scratch_bytes(M, K1, 32))
        stride = (nb + 255) // 256 * 256
        slabs = torch.zeros(batch * stride, dtype=torch.uint8, device=DEV)
        ldg = (M + 31) // 32 * 32
        for b in range(batch):
            X = rnd(M, K1, seed=10 * seed + b)
            Gt = torch.zeros(32, ldg, dtype=torch.bfloat16, device=DEV)
            Gt[:, :M] = rnd(M, 32, seed=20 * seed + b, scale=0.5).t()
            L().check(lib.cara_tskinny_partial(p(X), K1, p(Gt), ldg, C.c_void_p(slabs.data_ptr() + b * stride), 1 if want_cs else 0,
                                               M, K1, 32, st()), "partial")
        D1, D2 = (torch.full((batch, K1, 32), float("nan"), device=DEV) for _ in range(2))
        c1, c2 = (torch.full((batch, K1), float("nan"), device=DEV) for _ in range(2))
        L().check(lib.cara_tskinny_reduce(p(slabs), C.c_size_t(stride), p(D1), p(c1) if want_cs else None, batch, M, K1, 32, st()), "reduce")
        probs.append(Red(slabs.data_ptr(), stride, D2.data_ptr(), c2.data_ptr() if want_cs else None, batch, M, K1, 32))
        keep.append((D1, D2, c1, c2, want_cs, slabs))
    arr = (Red * len(probs))(*probs)
    L().check(lib.cara_tskinny_reduce_many(arr, len(probs), st()), "reduce_many")
    for D1, D2, c1, c2, want_cs, _ in keep:
        assert torch.equal(D1, D2) and (not want_cs or torch.equal(c1, c2))
    assert lib.cara_tskinny_reduce_many(arr, 17, st()) != 0      # more than CARA_TS_REDUCE_MAX entries


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C_", [(12608, 768), (64, 768), (33, 1024)])
def test_layernorm_fwd_bwd(M, C_):
    lib = L().lib()
    x = rnd(M, C_, seed=1, scale=2.0, dtype=torch.float32) + 0.5
    g = 1 + 0.1 * rnd(C_, seed=2, dtype=torch.float32)
    b = 0.1 * rnd(C_, seed=3, dtype=torch.float32)
    y = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    p, st = L().ptr, L().stream
    L().check(lib.cara_layernorm_fwd(p(x), C.c_long(C_), p(g), p(b), p(y), p(mean), p(rstd), M, C_, C.c_float(1e-6), st()), "ln fwd")
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (C_,), g.double(), b.double(), 1e-6)
    close(y, ref, 2 ** -8, 1e-3, "ln fwd")
    close(mean, x.double().mean(1), 1e-5, 1e-5, "ln mean")
    dy = rnd(M, C_, seed=4)
    ref.backward(dy.double())
    dx_in = rnd(M, C_, seed=5, dtype=torch.float32)
    dx = torch.empty(M, C_, device=DEV)
    dyb = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    rps = 7
    rs = (torch.arange((M + rps - 1) // rps, device=DEV) % 3).float() * 0.5
    L().check(lib.cara_layernorm_bwd(p(dy), p(x), C.c_long(C_), p(g), p(mean), p(rstd), p(dx_in), p(dx), p(dyb), p(rs),
                                     rps, M, C_, st()), "ln bwd")
    refdx = dx_in.double() + xd.grad
    close(dx, refdx, 1e-4, 1e-4, "ln bwd dx")
    close(dyb, refdx * rs.double().repeat_interleave(rps)[:M, None], 2 ** -8, 1e-3, "ln bwd dyb")


@pytest.mark.parametrize("M,C_,rank,Rp", [(12608, 768, 16, 32), (333, 768, 8, 32), (197, 1024, 16, 32), (70, 768, 32, 32), (5, 256, 3, 32),
                                          (12608, 768, 64, 64), (333, 1024, 40, 64), (21, 256, 64, 64)])
def test_layernorm_fused_adapter_contraction(M, C_, rank, Rp):
    """cara_layernorm_fwd_xu / _bwd_xu: LayerNorm + the skinny product of the next linear (T = y U, G' = dyb Vs)
    from the row held in registers; must agree with the separate cara_skinny_xu pass on the same bf16 rows."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    x = rnd(M, C_, seed=1, scale=2.0, dtype=torch.float32) + 0.5
    g = 1 + 0.1 * rnd(C_, seed=2, dtype=torch.float32)
    b = 0.1 * rnd(C_, seed=3, dtype=torch.float32)
    Ut = rnd(Rp, C_, seed=6, scale=0.1)
    Ut[rank:] = 0                                   # pack rows beyond the rank are zero
    ldt = (M + 31) // 32 * 32
    y = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    T = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Tt = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_layernorm_fwd_xu(p(x), C.c_long(C_), p(g), p(b), p(y), p(mean), p(rstd), M, C_, C.c_float(1e-6),
                                        p(Ut), rank, Rp, p(T), p(Tt), ldt, st()), "ln fwd xu")
    y2 = torch.empty_like(y)
    L().check(lib.cara_layernorm_fwd(p(x), C.c_long(C_), p(g), p(b), p(y2), p(mean), p(rstd), M, C_, C.c_float(1e-6), st()), "ln fwd")
    assert torch.equal(y, y2)
    ref = y.double() @ Ut.double().t()
    close(T, ref, 2 ** -8, 1e-3, "fused T")
    assert torch.equal(Tt[:, :M], T.t()) and torch.count_nonzero(Tt[:, M:]) == 0
    assert torch.count_nonzero(T[:, rank:]) == 0
    T2 = torch.empty_like(T)
    L().skinny_xu(y, Ut, T2)
    close(T, T2.double(), 2 ** -7, 1e-3, "fused T vs cara_skinny_xu")   # two roundings of nearly equal fp32 sums: <= 1 bf16 ulp
    # backward: G' from the row-scaled bf16 gradient the kernel emits
    dy = rnd(M, C_, seed=4)
    dx_in = rnd(M, C_, seed=5, dtype=torch.float32)
    dx, dx2 = torch.empty(M, C_, device=DEV), torch.empty(M, C_, device=DEV)
    dyb, dyb2 = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV), torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    rps = 7
    rs = (torch.arange((M + rps - 1) // rps, device=DEV) % 3).float() * 0.5 + 0.5
    G = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Gt = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_layernorm_bwd_xu(p(dy), p(x), C.c_long(C_), p(g), p(mean), p(rstd), p(dx_in), p(dx), p(dyb), p(rs),
                                        rps, M, C_, p(Ut), rank, Rp, p(G), p(Gt), ldt, st()), "ln bwd xu")
    L().check(lib.cara_layernorm_bwd(p(dy), p(x), C.c_long(C_), p(g), p(mean), p(rstd), p(dx_in), p(dx2), p(dyb2), p(rs),
                                     rps, M, C_, st()), "ln bwd")
    assert torch.equal(dx, dx2) and torch.equal(dyb, dyb2)
    close(G, dyb.double() @ Ut.double().t(), 2 ** -8, 1e-3, "fused G'")
    assert torch.equal(Gt[:, :M], G.t()) and torch.count_nonzero(Gt[:, M:]) == 0


@pytest.mark.parametrize("M,C_,fused", [(12608, 768, True), (333, 768, False), (197, 1024, True), (70, 256, False)])
def test_layernorm_outputs_in_k_panel_major_layout(M, C_, fused):
    """cara_layernorm_fwd_ex / _bwd_ex with y_panels / dyb_panels = M: the bf16 outputs land as [C/32][M][32] panels,
    bitwise the row-major values; statistics, dx and the fused skinny products are untouched."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    x = rnd(M, C_, seed=1, scale=2.0, dtype=torch.float32) + 0.5
    g = 1 + 0.1 * rnd(C_, seed=2, dtype=torch.float32)
    b = 0.1 * rnd(C_, seed=3, dtype=torch.float32)
    Ut = rnd(32, C_, seed=6, scale=0.1)
    ldt = (M + 31) // 32 * 32
    up = p(Ut) if fused else None
    y1 = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    y2 = torch.full((C_ // 32, M, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    mean, rstd, mean2, rstd2 = (torch.empty(M, device=DEV) for _ in range(4))
    T1, T2 = (torch.zeros(M, 32, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    Tt1, Tt2 = (torch.zeros(32, ldt, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    L().check(lib.cara_layernorm_fwd_ex(p(x), C.c_long(C_), p(g), p(b), p(y1), p(mean), p(rstd), M, C_, C.c_float(1e-6),
                                        up, 32, 32, p(T1), p(Tt1), ldt, 0, st()), "ln fwd ex")
    L().check(lib.cara_layernorm_fwd_ex(p(x), C.c_long(C_), p(g), p(b), p(y2), p(mean2), p(rstd2), M, C_, C.c_float(1e-6),
                                        up, 32, 32, p(T2), p(Tt2), ldt, M, st()), "ln fwd ex panels")
    assert torch.equal(y2, _panels(y1)) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert torch.equal(T1, T2) and torch.equal(Tt1, Tt2)
    dy = rnd(M, C_, seed=4)
    dx_in = rnd(M, C_, seed=5, dtype=torch.float32)
    dx1, dx2 = torch.empty(M, C_, device=DEV), torch.empty(M, C_, device=DEV)
    d1 = torch.empty(M, C_, dtype=torch.bfloat16, device=DEV)
    d2 = torch.full((C_ // 32, M, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    rps = 7
    rs = (torch.arange((M + rps - 1) // rps, device=DEV) % 3).float() * 0.5 + 0.5
    L().check(lib.cara_layernorm_bwd_ex(p(dy), p(x), C.c_long(C_), p(g), p(mean), p(rstd), p(dx_in), p(dx1), p(d1), p(rs),
                                        rps, M, C_, up, 32, 32, p(T1), p(Tt1), ldt, 0, st()), "ln bwd ex")
    L().check(lib.cara_layernorm_bwd_ex(p(dy), p(x), C.c_long(C_), p(g), p(mean), p(rstd), p(dx_in), p(dx2), p(d2), p(rs),
                                        rps, M, C_, up, 32, 32, p(T2), p(Tt2), ldt, M, st()), "ln bwd ex panels")
    assert torch.equal(d2, _panels(d1)) and torch.equal(dx1, dx2) and torch.equal(T1, T2) and torch.equal(Tt1, Tt2)


def test_layernorm_strided_cls_rows():
    """Final norm: only the cls row of each sample (row stride tokens*C)."""
    lib = L().lib()
    B, T, C_ = 5, 7, 768
    x = rnd(B * T, C_, seed=1, dtype=torch.float32)
    g, b = 1 + 0.1 * rnd(C_, seed=2, dtype=torch.float32), 0.1 * rnd(C_, seed=3, dtype=torch.float32)
    y = torch.empty(B, C_, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    p, st = L().ptr, L().stream
    L().check(lib.cara_layernorm_fwd(p(x), C.c_long(T * C_), p(g), p(b), p(y), p(mean), p(rstd), B, C_, C.c_float(1e-6), st()), "ln")
    ref = torch.nn.functional.layer_norm(x.double().reshape(B, T, C_)[:, 0], (C_,), g.double(), b.double(), 1e-6)
    close(y, ref, 2 ** -8, 1e-3, "ln strided")


# ------------------------------------------------------------------------------------------
def attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B * N, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 5, 2), (1, 64, 1), (2, 224, 3), (1, 33, 2),
                                   # the persistent kernels (128 < N <= 224) at every tile count and ragged last tiles
                                   (2, 129, 2), (1, 160, 3), (2, 161, 2), (1, 193, 1), (30, 197, 12),
                                   # above 224 tokens: two-sweep forward, key groups on grid.y (ViT-L/16 @384 has 577)
                                   (2, 577, 16), (1, 225, 2), (1, 608, 1), (2, 300, 3)])
def test_attention_fwd_bwd(B, N, H):
    lib = L().lib()
    scale = 64 ** -0.5
    qkv = rnd(B * N, 3 * H * 64, seed=1, scale=1.0)
    out = torch.full((B * N, H * 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    p, st = L().ptr, L().stream
    L().check(lib.cara_attention_fwd(p(qkv), p(out), p(lse), B, N, H, C.c_float(scale), st()), "attn fwd")
    qd = qkv.double().requires_grad_(True)
    ref, ref_lse = attn_ref(qd, B, N, H, scale)
    # P is rounded to bf16 before P.V (2^-9 relative on each weight), output rounded to bf16
    close(out, ref, 2 ** -7, 4e-3, "attn out")
    close(lse, ref_lse, 1e-4, 1e-4, "attn lse")
    dout = rnd(B * N, H * 64, seed=2)
    ref.backward(dout.double())
    dqkv = torch.full_like(qkv, float("nan"))
    L().check(lib.cara_attention_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), B, N, H, C.c_float(scale), st()), "attn bwd")
    g = qd.grad
    err = (dqkv.double() - g).abs()
    tol = 2 ** -6 * g.abs() + 0.02 * g.abs().max()
    assert not torch.isnan(dqkv).any()
    assert (err <= tol).all(), f"attn bwd max err {err.max():.3e} vs grad max {g.abs().max():.3e}"
    rel = (dqkv.double() - g).norm() / g.norm()
    assert rel < 8e-3, f"attn bwd rel-L2 {rel:.3e}"


@pytest.mark.parametrize("N", [197, 577])
def test_attention_softmax_extremes(N):
    """Large-magnitude scores: exact two-pass softmax must not overflow."""
    lib = L().lib()
    B, H = 1, 1
    qkv = rnd(B * N, 3 * 64, seed=3, scale=6.0)
    out = torch.empty(B * N, 64, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    L().check(lib.cara_attention_fwd(L().ptr(qkv), L().ptr(out), L().ptr(lse), B, N, H, C.c_float(0.125), L().stream()), "attn")
    ref, _ = attn_ref(qkv, B, N, H, 0.125)
    assert torch.isfinite(out).all()
    close(out, ref, 2 ** -6, 0.05, "attn extreme")


# ------------------------------------------------------------------------------------------
def test_im2col_assemble_xent_transpose():
    lib = L().lib()
    p, st = L().ptr, L().stream
    B, Cc, Hi, P = 3, 3, 64, 16
    img = rnd(B, Cc, Hi, Hi, seed=1, dtype=torch.float32)
    gh = Hi // P
    patches = torch.empty(B * gh * gh, Cc * P * P, dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_im2col_patches(p(img), p(patches), B, Cc, Hi, Hi, P, st()), "im2col")
    ref = img.reshape(B, Cc, gh, P, gh, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gh, -1).bfloat16()
    assert torch.equal(patches, ref)
    D = 768
    emb = rnd(B * gh * gh, D, seed=2, dtype=torch.float32)
    cls = rnd(D, seed=3, dtype=torch.float32)
    pos = rnd(gh * gh + 1, D, seed=4, dtype=torch.float32)
    x = torch.empty(B, gh * gh + 1, D, device=DEV)
    L().check(lib.cara_assemble_tokens(p(emb), p(cls), p(pos), p(x), B, gh * gh, D, st()), "assemble")
    refx = torch.cat((cls.expand(B, 1, D), emb.reshape(B, gh * gh, D)), 1) + pos
    assert torch.equal(x, refx)
    Bc, Cn = 64, 100
    logits = rnd(Bc, Cn, seed=5, scale=3.0, dtype=torch.float32)
    labels = torch.randint(0, Cn, (Bc,), generator=torch.Generator().manual_seed(6)).to(DEV)
    loss = torch.empty(1 + Bc, device=DEV)
    dl = torch.empty(Bc, Cn, device=DEV)
    L().check(lib.cara_cross_entropy(p(logits), p(labels), p(loss), p(dl), Bc, Cn, st()), "xent")
    ld = logits.double().requires_grad_(True)
    rl = torch.nn.functional.cross_entropy(ld, labels)
    rl.backward()
    close(loss[0], rl.detach(), 1e-5, 1e-6, "xent loss")
    close(dl, ld.grad, 1e-4, 1e-7, "xent grad")
    src = rnd(100, 70, seed=7)
    dst = torch.empty(70, 100, dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_transpose_bf16(p(src), p(dst), 100, 70, st()), "transpose")
    assert torch.equal(dst, src.t().contiguous())
    f = rnd(1001, seed=8, dtype=torch.float32)
    o = torch.empty(1001, dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_f32_to_bf16(p(f), p(o), C.c_size_t(1001), st()), "cvt")
    assert torch.equal(o, f.bfloat16())


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rank,Rp", [(16, 32), (8, 32), (32, 32), (64, 64)])
def test_factor_prep_and_grad_reduce(rank, Rp):
    """cara_factor_prep vs the oracle's A.3 table; cara_factor_grad_reduce vs autograd through it."""
    from oracle import cara_oracle as O
    lib = L().lib()
    p, st = L().ptr, L().stream
    depth, dim, heads, s = 12, 768, 12, 0.1
    cp = O.synthetic_cp(rank=rank)
    cpd = {k: v.to(DEV).contiguous() for k, v in cp.items()}
    geom = L().Geom(depth, dim, heads, rank, Rp, s)
    lay = L().PackLayout()
    L().check(lib.cara_pack_offsets(C.byref(geom), C.byref(lay)), "offsets")
    pack = torch.zeros(lay.total, dtype=torch.uint8, device=DEV)
    cps = L().CpPtrs(*[p(cpd["CP_" + n]) for n in L().CP_FIELDS])
    bb = [rnd(depth, n, seed=i, dtype=torch.float32) for i, n in ((1, dim), (2, 4 * dim), (3, dim))]
    L().check(lib.cara_factor_prep(C.byref(geom), C.byref(cps), p(bb[0]), p(bb[1]), p(bb[2]), p(pack), st()), "prep")
    torch.cuda.synchronize()
    fac = O.build_factored(cp, s)

    def view(l, off, rows, cols, dt=torch.bfloat16):
        n = rows * cols * (2 if dt == torch.bfloat16 else 4)
        return pack[l * lay.layer_stride + off: l * lay.layer_stride + off + n].view(dt).reshape(rows, cols).cpu()

    for l in (0, 5, 11):
        for name, din, dout in (("qkv", dim, 3 * dim), ("proj", dim, dim), ("fc1", dim, 4 * dim), ("fc2", 4 * dim, dim)):
            U, Vs, cs = fac[l][name]
            Upad = torch.zeros(din, Rp); Upad[:, :rank] = U
            Vpad = torch.zeros(dout, Rp); Vpad[:, :rank] = Vs
            assert torch.equal(view(l, getattr(lay, "U_" + name), din, Rp), Upad.bfloat16()), (l, name, "U")
            assert torch.equal(view(l, getattr(lay, "Ut_" + name), Rp, din), Upad.t().bfloat16()), (l, name, "Ut")
            got = view(l, getattr(lay, "Vs_" + name), dout, Rp).float()
            assert torch.allclose(got, Vpad.bfloat16().float(), rtol=2 ** -7, atol=1e-8), (l, name, "Vs")
            assert torch.equal(view(l, getattr(lay, "Vst_" + name), Rp, dout).float(), got.t()), (l, name, "Vst")
        for i, (nm, key) in enumerate((("bias_proj", "CP_bias1"), ("bias_fc1", "CP_bias2"), ("bias_fc2", "CP_bias3"))):
            n = cp[key].numel()
            ref = bb[i][l].cpu() + s * cp[key]
            assert torch.allclose(view(l, getattr(lay, nm), 1, n, torch.float32)[0], ref, rtol=1e-6, atol=1e-7)

    # gradient scatter: random per-layer dU/dVs/dc, compare with autograd through build_factored
    g = torch.Generator().manual_seed(9)
    names = (("qkv", dim, 3 * dim), ("proj", dim, dim), ("fc1", dim, 4 * dim), ("fc2", 4 * dim, dim))
    dU = {n: torch.randn(depth, di, Rp, generator=g) for n, di, do in names}
    dV = {n: torch.randn(depth, do, Rp, generator=g) for n, di, do in names}
    dc = {n: torch.randn(depth, do, generator=g) for n, di, do in names if n != "qkv"}
    cpv = {k: v.double().clone().requires_grad_(True) for k, v in cp.items()}
    facv = O.build_factored(cpv, s)
    tot = 0
    for l in range(depth):
        for n, di, do in names:
            U, Vs, cs = facv[l][n]
            tot = tot + (U * dU[n][l, :, :rank].double()).sum() + (Vs * dV[n][l, :, :rank].double()).sum()
            if cs is not None:
                tot = tot + (cs * dc[n][l].double()).sum()
    tot.backward()
    dev = lambda t: t.to(DEV).contiguous()  # noqa: E731
    keep = [dev(dU["qkv"]), dev(dV["qkv"]), dev(dU["proj"]), dev(dV["proj"]), dev(dU["fc1"]), dev(dV["fc1"]),
            dev(dU["fc2"]), dev(dV["fc2"]), dev(dc["proj"]), dev(dc["fc1"]), dev(dc["fc2"])]
    lg = L().LayerGrads(*[p(t) for t in keep])
    gout = {n: torch.full_like(cpd["CP_" + n], float("nan")) for n in L().CP_FIELDS}
    gp = L().CpPtrs(*[p(gout[n]) for n in L().CP_FIELDS])
    scratch = torch.empty(lib.cara_factor_grad_scratch_bytes(C.byref(geom)), dtype=torch.uint8, device=DEV)
    L().check(lib.cara_factor_grad_reduce(C.byref(geom), C.byref(cps), C.byref(lg), C.byref(gp), p(scratch), st()), "grad reduce")
    torch.cuda.synchronize()
    for n in L().CP_FIELDS:
        ref = cpv["CP_" + n].grad
        scale_ = max(ref.abs().max().item(), 1e-6)
        close(gout[n], ref, 1e-4, 1e-4 * scale_, "grad " + n)


def test_transposing_lds_read_semantics():
    """ds_read_b64_tr_b16 as the attention kernels use it: (a) raw lane semantics over sm[i] = i
    (lane 4q+p of a 16-lane group supplies row q / columns 4p..4p+3 of a 4x16 block, lane i receives
    column i of the four rows), (b) the kernels' own staging + fragment helper on a real matrix."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    ad = []
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        ad.append(2 * (g * 1024 + (i >> 2) * 64 + 4 * (i & 3)))
    a = torch.tensor(ad, dtype=torch.int32, device=DEV)
    o = torch.empty(256, dtype=torch.int16, device=DEV)
    L().check(lib.cara_debug_tr_probe(p(a), p(o), st()), "tr probe")
    got = o.cpu().reshape(64, 4)
    exp = torch.tensor([[(l >> 4) * 1024 + q * 64 + (l & 15) for q in range(4)] for l in range(64)], dtype=torch.int16)
    assert torch.equal(got, exp)
    N = 197
    V = (torch.arange(N).reshape(N, 1) % 16 * 16 + torch.arange(64).reshape(1, 64) % 16).float() + (torch.arange(64).reshape(1, 64) // 16) * 0.25
    Vb = V.bfloat16().to(DEV).contiguous()
    for cbase, base in ((0, 0), (32, 16), (0, 176), (32, 208)):
        out = torch.empty(64, 8, dtype=torch.bfloat16, device=DEV)
        L().check(lib.cara_debug_tr_frag(p(Vb), p(out), N, cbase, base, st()), "tr frag")
        rows = torch.tensor([[min(base + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3), N - 1) for j in range(8)] for l in range(64)])
        cols = torch.tensor([[cbase + (l & 31)] * 8 for l in range(64)])
        assert torch.equal(out.cpu(), Vb.cpu()[rows, cols]), (cbase, base)


@pytest.mark.parametrize("dim,depth,rank", [(256, 2, 4), (768, 3, 16), (128, 1, 64)])
def test_dense_delta_entry_points(dim, depth, rank):
    """Order-2 QKV tensorisation (dim_experiment.py:203-207): cara_dense_delta_materialize against
    s * sum_r R1 A1[3l+k] A2[e dim + o] in fp64 (bf16 outputs: half an ulp), and cara_dense_delta_grad against fp64 autograd of
    sum(dD * tensor) through CP_A1 / CP_A2 / CP_R1; cara_sum_slabs_f32 bit-exact against the same fixed-order sum."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    s = 0.3
    g = torch.Generator().manual_seed(dim + rank)
    A1 = torch.randn(3 * depth, rank, generator=g)
    A2 = 0.1 * torch.randn(dim * dim, rank, generator=g)
    R1 = 1.0 + 0.2 * torch.randn(rank, generator=g)
    dD = torch.randn(depth, 3, dim, dim, generator=g)
    d = {n: t.to(DEV).contiguous() for n, t in (("A1", A1), ("A2", A2), ("R1", R1))}
    geom = L().Geom(depth, dim, 4, rank, 64 if rank > 32 else 32, s, 2)
    cps = L().CpPtrs(A1=p(d["A1"]), A2=p(d["A2"]), R1=p(d["R1"]))
    Dm = torch.empty(depth, 3 * dim, dim, dtype=torch.bfloat16, device=DEV)
    Dmt = torch.empty(depth, dim, 3 * dim, dtype=torch.bfloat16, device=DEV)
    L().check(lib.cara_dense_delta_materialize(C.byref(geom), C.byref(cps), p(Dm), p(Dmt), st()), "materialize")
    v = {n: t.double().requires_grad_(True) for n, t in (("A1", A1), ("A2", A2), ("R1", R1))}
    ten = s * torch.einsum("r,lkr,pr->lkp", v["R1"], v["A1"].reshape(depth, 3, rank), v["A2"]).reshape(depth, 3, dim, dim)   # [l][k][e][o]
    want = ten.detach().permute(0, 1, 3, 2).reshape(depth, 3 * dim, dim)      # [l][k dim + o][e]
    assert (Dm.double().cpu() - want).abs().max() <= 2.0 ** -8 * want.abs().max()
    assert torch.equal(Dmt, Dm.transpose(1, 2).contiguous())
    (ten * dD.double()).sum().backward()
    gout = {n: torch.full_like(d[n], float("nan")) for n in d}
    gp = L().CpPtrs(A1=p(gout["A1"]), A2=p(gout["A2"]), R1=p(gout["R1"]))
    nb = lib.cara_dense_delta_grad_scratch_bytes(C.byref(geom))
    assert nb > 0
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dDd = dD.to(DEV)
    L().check(lib.cara_dense_delta_grad(C.byref(geom), C.byref(cps), p(dDd), C.byref(gp), p(scratch), st()), "dense delta grad")
    torch.cuda.synchronize()
    for n in d:
        ref = v[n].grad
        close(gout[n], ref, 1e-4, 1e-4 * ref.abs().max().item(), "order-2 grad " + n)
    # refusals: another order's geometry, a missing tensor
    g4 = L().Geom(depth, dim, 4, rank, 32, s, 4)
    assert lib.cara_dense_delta_materialize(C.byref(g4), C.byref(cps), p(Dm), p(Dmt), st()) != 0
    assert lib.cara_dense_delta_grad_scratch_bytes(C.byref(g4)) == 0
    assert lib.cara_dense_delta_grad(C.byref(geom), C.byref(cps), None, C.byref(gp), p(scratch), st()) != 0
    # slabs
    slabs = torch.randn(5, 1000, device=DEV)
    out = torch.empty(777, device=DEV)
    L().check(lib.cara_sum_slabs_f32(p(slabs), 5, C.c_size_t(1000), C.c_size_t(777), p(out), st()), "sum slabs")
    ref = ((slabs[0] + slabs[2]) + slabs[4]) + (slabs[1] + slabs[3])
    assert torch.equal(out, ref[:777])
    assert lib.cara_sum_slabs_f32(p(slabs), 0, C.c_size_t(1000), C.c_size_t(777), p(out), st()) != 0


def test_adamw_step_against_torch():
    """cara_amd.optim.AdamW (one cara_adamw_step launch over all tensors) against torch.optim.AdamW on the CPU: five steps, two
    parameter groups with their own lr / weight decay, an lr that a scheduler changes between steps, sizes on both sides of the
    1024-element chunk; then torch's state_dict layout and a reload."""
    from cara_amd.optim import AdamW
    g = torch.Generator().manual_seed(21)
    shapes = [(36, 16), (768, 16), (16,), (3072,), (100, 768), (1,), (1025,), (1024,)]
    ref = [torch.randn(*s_, generator=g).requires_grad_(True) for s_ in shapes]
    dev = [r.detach().clone().to(DEV).requires_grad_(True) for r in ref]
    groups = lambda ps: [{"params": ps[:5], "lr": 1e-3, "weight_decay": 1e-4}, {"params": ps[5:], "lr": 3e-3, "weight_decay": 0.0}]  # noqa: E731
    ropt = torch.optim.AdamW(groups(ref), betas=(0.9, 0.999), eps=1e-8)
    dopt = AdamW(groups(dev), betas=(0.9, 0.999), eps=1e-8)
    for step in range(5):
        for r, d in zip(ref, dev):
            r.grad = torch.randn(r.shape, generator=g) * (10.0 ** (step - 2))
            d.grad = r.grad.to(DEV)
        for o in (ropt, dopt):
            o.param_groups[0]["lr"] = 1e-3 * (0.5 ** step)
        ropt.step()
        dopt.step()
        for r, d in zip(ref, dev):
            close(d.detach(), r.detach(), 2e-6, 2e-7, f"AdamW step {step}")
    sd = dopt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    close(sd["state"][4]["exp_avg_sq"], ropt.state_dict()["state"][4]["exp_avg_sq"], 2e-6, 1e-12, "exp_avg_sq")
    d2 = AdamW(groups(dev))
    d2.load_state_dict(sd)
    for r, d in zip(ref, dev):
        r.grad = torch.ones_like(r)
        d.grad = torch.ones_like(d)
    ropt.step()
    d2.step()
    for r, d in zip(ref, dev):
        close(d.detach(), r.detach(), 2e-6, 2e-7, "AdamW after reload")
    # no CPU path
    c = torch.zeros(4, requires_grad=True)
    c.grad = torch.ones(4)
    with pytest.raises(L().CaraError, match="GPU"):
        AdamW([c]).step()


@pytest.mark.parametrize("M,K,rank", [(12608, 3072, 16), (12608, 768, 16), (333, 2304, 5), (64, 3072, 16), (12608, 768, 17), (70, 768, 32)])
def test_skinny_xu_with_the_rank_stated(M, K, rank):
    """cara_skinny_xu_r: at Rp = 32 and rank <= 16 only the first column tile is computed and the rest is WRITTEN as zeros (the
    outputs start as NaN here); bit for bit the result of cara_skinny_xu, transposed copy included."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    X = rnd(M, K, seed=1)
    Ut = rnd(32, K, seed=2, scale=0.1)
    Ut[rank:] = 0
    ldt = (M + 31) // 32 * 32
    out = []
    for fn, extra in ((lib.cara_skinny_xu, ()), (lib.cara_skinny_xu_r, (rank,))):
        T = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
        Tt = torch.full((32, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().check(fn(p(X), K, p(Ut), p(T), p(Tt), ldt, M, K, 32, *extra, st()), "skinny")
        out.append((T, Tt))
    (T0, Tt0), (T1, Tt1) = out
    assert torch.equal(T0, T1) and torch.equal(Tt0[:, :M], Tt1[:, :M])
    assert torch.count_nonzero(T1[:, rank:]) == 0 and torch.equal(Tt1[:, :M], T1.t())
    close(T1, X.double() @ Ut.double().t(), 2 ** -8, 1e-3, "T")
    assert lib.cara_skinny_xu_r(p(X), K, p(Ut), p(T), p(Tt), ldt, M, K, 32, 0, st()) != 0
    assert lib.cara_skinny_xu_r(p(X), K, p(Ut), p(T), p(Tt), ldt, M, K, 32, 33, st()) != 0


@pytest.mark.parametrize("B,N,H", [(3, 197, 12), (2, 5, 2), (1, 577, 16), (64, 197, 12)])
def test_attention_for_the_cls_query_alone(B, N, H):
    """cara_attention_cls_fwd / _bwd (what the last block needs: only the cls row of its attention output reaches the logits)
    against the full kernels' cls rows and against fp64; the backward with a gradient on the cls rows only, all of dqkv."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    scale = 64 ** -0.5
    qkv = rnd(B * N, 3 * H * 64, seed=1, scale=1.0)
    out_full = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=DEV)
    lse_full = torch.empty(B, H, N, device=DEV)
    L().check(lib.cara_attention_fwd(p(qkv), p(out_full), p(lse_full), B, N, H, C.c_float(scale), st()), "attn fwd")
    out = torch.full((B * N, H * 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.full((B, H, N), float("nan"), device=DEV)
    L().check(lib.cara_attention_cls_fwd(p(qkv), p(out), p(lse), B, N, H, C.c_float(scale), st()), "attn cls fwd")
    cls = torch.arange(B, device=DEV) * N
    qd = qkv.double().requires_grad_(True)
    ref, ref_lse = attn_ref(qd, B, N, H, scale)
    close(out[cls], ref[cls], 2 ** -7, 4e-3, "cls out")
    close(out[cls], out_full[cls].double(), 2 ** -7, 2e-3, "cls out vs the full kernel")
    close(lse[:, :, 0], ref_lse[:, :, 0], 1e-4, 1e-4, "cls lse")
    mask = torch.ones(B * N, dtype=torch.bool, device=DEV)
    mask[cls] = False
    assert torch.isnan(out[mask]).all() and torch.isnan(lse[:, :, 1:]).all()          # nothing else is written
    dout = torch.zeros(B * N, H * 64, dtype=torch.bfloat16, device=DEV)
    dout[cls] = rnd(B, H * 64, seed=2)
    ref.backward(dout.double())
    dqkv = torch.full_like(qkv, float("nan"))
    L().check(lib.cara_attention_cls_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), B, N, H, C.c_float(scale), st()), "attn cls bwd")
    g = qd.grad
    assert not torch.isnan(dqkv).any()
    err = (dqkv.double() - g).abs()
    assert (err <= 2 ** -6 * g.abs() + 0.02 * g.abs().max()).all(), f"max err {err.max():.3e} vs grad max {g.abs().max():.3e}"
    assert (dqkv.double() - g).norm() / g.norm() < 8e-3
    if N > 1:
        qblock = dqkv.reshape(B, N, 3, H * 64)[:, 1:, 0]
        assert torch.count_nonzero(qblock) == 0                                          # no query but the cls one was in play
    dfull = torch.empty_like(qkv)
    L().check(lib.cara_attention_bwd(p(qkv), p(out_full), p(dout), p(lse_full), p(dfull), B, N, H, C.c_float(scale), st()), "attn bwd")
    assert (dqkv.double() - dfull.double()).norm() / dfull.double().norm() < 8e-3


@pytest.mark.parametrize("M,K1a,K1b,rank", [(12608, 768, 3072, 16), (333, 768, 2304, 7), (64, 3072, 768, 16)])
def test_transposed_skinny_products_with_the_rank_stated(M, K1a, K1b, rank):
    """cara_tskinny_partial2_r + cara_tskinny_reduce_many with Rc = 16: at Rp = 32 and rank <= 16 the products compute 16 of their
    32 columns into 16-wide slabs; the reduced D equals the all-columns path bit for bit on those columns, the rest is written
    as zeros; column sums unchanged."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    Rp = 32
    ldg = (M + 31) // 32 * 32
    Xa, Xb = rnd(M, K1a, seed=1), rnd(M, K1b, seed=2)
    Ga = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Gb = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Ga[:rank, :M] = rnd(rank, M, seed=3)
    Gb[:rank, :M] = rnd(rank, M, seed=4)
    res = []
    for half in (False, True):
        sa = torch.empty(lib.cara_tskinny_scratch_bytes(M, K1a, Rp), dtype=torch.uint8, device=DEV)
        sb = torch.empty(lib.cara_tskinny_scratch_bytes(M, K1b, Rp), dtype=torch.uint8, device=DEV)
        Da = torch.full((K1a, Rp), float("nan"), device=DEV)
        Db = torch.full((K1b, Rp), float("nan"), device=DEV)
        cs = torch.full((K1b,), float("nan"), device=DEV)
        if half:
            L().check(lib.cara_tskinny_partial2_r(p(Xa), K1a, p(Ga), p(sa), K1a, p(Xb), K1b, p(Gb), p(sb), K1b, 1, ldg, M, Rp, rank, st()), "partial2_r")
        else:
            L().check(lib.cara_tskinny_partial2(p(Xa), K1a, p(Ga), p(sa), K1a, p(Xb), K1b, p(Gb), p(sb), K1b, 1, ldg, M, Rp, st()), "partial2")
        tab = (L().TsReduce * 2)(L().TsReduce(p(sa), 0, p(Da), None, 1, M, K1a, Rp, 16 if half else 0),
                                 L().TsReduce(p(sb), 0, p(Db), p(cs), 1, M, K1b, Rp, 16 if half else 0))
        L().check(lib.cara_tskinny_reduce_many(tab, 2, st()), "reduce many")
        torch.cuda.synchronize()
        res.append((Da, Db, cs))
    (Da0, Db0, cs0), (Da1, Db1, cs1) = res
    assert torch.equal(Da0, Da1) and torch.equal(Db0, Db1) and torch.equal(cs0, cs1)
    assert torch.count_nonzero(Da1[:, rank:]) == 0 and torch.count_nonzero(Db1[:, rank:]) == 0
    close(Da1[:, :rank], Xa.double().t() @ Ga[:rank, :M].double().t(), 1e-3, 1e-2, "dU")
    close(cs1, Xb.double().sum(0), 1e-3, 1e-2, "colsum")
    bad = (L().TsReduce * 1)(L().TsReduce(p(sa), 0, p(Da), None, 1, M, K1a, Rp, 8))
    assert lib.cara_tskinny_reduce_many(bad, 1, st()) != 0


@pytest.mark.parametrize("M,N,K,rank,epi", [(12608, 768, 3072, 16, "resid"), (12608, 768, 768, 16, "resid"), (333, 768, 768, 5, "bf16"),
                                            (12608, 768, 3072, 17, "resid")])
def test_adapter_inside_gemm_with_the_rank_stated(M, N, K, rank, epi):
    """cara_gemm_args.Ut_rank: at Rp = 32 and rank <= 16 the adapter-inside GEMM computes 16 of the 32 columns of T (two T tiles
    per wave instead of four, 16 rows of Ut per K step) and writes the rest as zeros -- outputs, T and its transpose bit for
    bit those of the all-columns kernel."""
    Rp = 32
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    Ut, Vs = rnd(Rp, K, seed=3, scale=0.1), rnd(N, Rp, seed=4, scale=0.3)
    Ut[rank:] = 0
    Vs[:, rank:] = 0
    ldt = (M + 31) // 32 * 32
    bias = rnd(N, seed=5, dtype=torch.float32)
    res = []
    for r in (0, rank):
        T = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
        Tt = torch.full((Rp, ldt), float("nan"), dtype=torch.bfloat16, device=DEV)
        if epi == "resid":
            aux = rnd(M, N, seed=6, dtype=torch.float32)
            out = torch.full((M, N), float("nan"), device=DEV)
            L().gemm(A, W, out, epi=L().EPI_RESID, bias=bias, B2=Vs, aux=aux, Ut=Ut, T_out=T, Tt_out=Tt, Ut_rank=r)
        else:
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            L().gemm(A, W, out, epi=L().EPI_BF16, bias=bias, B2=Vs, Ut=Ut, T_out=T, Tt_out=Tt, Ut_rank=r)
        res.append((out, T, Tt[:, :M].clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert torch.count_nonzero(res[1][1][:, rank:]) == 0 and torch.equal(res[1][2], res[1][1].t())
    close(res[1][1], A.double() @ Ut.double().t(), 2 ** -8, 1e-3 * math.sqrt(K / 64), "T inside")


@pytest.mark.parametrize("M,din,dout,rank,epi", [(1000, 768, 2304, 16, "bf16"), (394, 3072, 768, 8, "resid"), (12608, 3072, 768, 16, "resid")])
def test_one_adapted_linear_per_call(M, din, dout, rank, epi):
    """cara_linear_fwd / cara_linear_bwd (SURVEY 8b's minimum export set): the adapter linear of cara.py:25-42 / :50-58 / :75-82 /
    :87-93 in factored form as ONE call each way, against the as-written arithmetic in fp64 -- y = x W^T + b + (x U) Vs^T and the
    gradients dX, dU = X^T (dY Vs), dVs = dY^T (X U), dc = colsum dY that cara_factor_grad_reduce takes."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    Rp = 32
    X, W = rnd(M, din, seed=1), rnd(dout, din, seed=2, scale=0.05)
    U, Vs = rnd(din, Rp, seed=3, scale=0.1), rnd(dout, Rp, seed=4, scale=0.3)
    U[:, rank:] = 0
    Vs[:, rank:] = 0
    bias = rnd(dout, seed=5, dtype=torch.float32)
    lin = L().Linear()
    lin.W, lin.Wt, lin.Ut, lin.U, lin.Vs, lin.Vst = p(W), p(W.t().contiguous()), p(U.t().contiguous()), p(U), p(Vs), p(Vs.t().contiguous())
    keep = [W.t().contiguous(), U.t().contiguous(), Vs.t().contiguous()]
    lin.W, lin.Wt, lin.Ut, lin.Vst = p(W), p(keep[0]), p(keep[1]), p(keep[2])
    lin.bias = p(bias)
    setattr(lin, "in", din)
    lin.out, lin.Rp, lin.rank = dout, Rp, rank
    ldt = (M + 31) // 32 * 32
    T = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Tt = torch.zeros(Rp, ldt, dtype=torch.bfloat16, device=DEV)
    if epi == "bf16":
        Y = torch.full((M, dout), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().check(lib.cara_linear_fwd(C.byref(lin), p(X), din, M, p(T), p(Tt), ldt, L().EPI_BF16, p(Y), 0, None, None, None, 0, st()), "cara_linear_fwd")
        ref = X.double() @ W.double().t() + bias.double() + T.double() @ Vs.double().t()
        close(Y, ref, 2 ** -8, 2e-3 * math.sqrt(din / 64), "linear forward")
    else:
        aux = rnd(M, dout, seed=6, dtype=torch.float32)
        rs = (torch.arange((M + 196) // 197, device=DEV) % 2).float() * 1.1
        Y = torch.full((M, dout), float("nan"), device=DEV)
        L().check(lib.cara_linear_fwd(C.byref(lin), p(X), din, M, p(T), p(Tt), ldt, L().EPI_RESID, p(Y), 0, None, p(aux), p(rs), 197, st()), "cara_linear_fwd")
        ref = aux.double() + rs.double().repeat_interleave(197)[:M, None] * (X.double() @ W.double().t() + bias.double() + T.double() @ Vs.double().t())
        close(Y, ref, 1e-5, 2e-3 * math.sqrt(din / 64), "linear forward (residual epilogue)")
    close(T, X.double() @ U.double(), 2 ** -8, 1e-3 * math.sqrt(din / 64), "T = X U")
    assert torch.equal(Tt[:, :M], T.t())
    # backward
    dY = rnd(M, dout, seed=7, scale=0.1)
    G = torch.full((M, Rp), float("nan"), dtype=torch.bfloat16, device=DEV)
    Gt = torch.zeros(Rp, ldt, dtype=torch.bfloat16, device=DEV)
    dX = torch.full((M, din), float("nan"), dtype=torch.bfloat16, device=DEV)
    su = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, din, Rp)), dtype=torch.uint8, device=DEV)
    sv = torch.zeros(int(lib.cara_tskinny_scratch_bytes(M, dout, Rp)), dtype=torch.uint8, device=DEV)
    dU, dVs, dc = torch.empty(din, Rp, device=DEV), torch.empty(dout, Rp, device=DEV), torch.empty(dout, device=DEV)
    L().check(lib.cara_linear_bwd(C.byref(lin), p(dY), dout, p(X), din, p(Tt), M, p(G), p(Gt), ldt, p(dX), 0, p(su), p(sv), p(dU), p(dVs), p(dc), st()),
              "cara_linear_bwd")
    close(G, dY.double() @ Vs.double(), 2 ** -8, 1e-3 * math.sqrt(dout / 64), "G' = dY Vs")
    close(dX, dY.double() @ W.double() + G.double() @ U.double().t(), 2 ** -8, 2e-3 * math.sqrt(dout / 64), "dX")
    close(dU, X.double().t() @ G.double(), 1e-3, 2e-2 * math.sqrt(M / 1000), "dU")
    close(dVs, dY.double().t() @ T.double(), 1e-3, 2e-2 * math.sqrt(M / 1000), "dVs")
    close(dc, dY.double().sum(0), 1e-3, 1e-2 * math.sqrt(M / 1000), "dc")
    bad = L().Linear()
    assert lib.cara_linear_fwd(C.byref(bad), p(X), din, M, p(T), p(Tt), ldt, L().EPI_BF16, p(Y), 0, None, None, None, 0, st()) != 0


def test_gelu_with_its_derivative_saved():
    """CARA_EPI_GELU_DG / CARA_EPI_MULH: the forward keeps gelu'(u) as IEEE half instead of u, the backward multiplies by it.  h is
    bitwise CARA_EPI_GELU's; the derivative against fp64 autograd at half precision; dH = acc * gelu' against fp64; the few-row and
    edge-tile paths; inference (C2 = NULL) leaves only h."""
    for (M, N, K) in ((12608, 3072, 768), (1500, 1024, 256), (64, 3072, 768), (333, 200, 128)):
        A, W, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3, dtype=torch.float32)
        scratch = torch.zeros(L().gemm_scratch_bytes(), dtype=torch.uint8, device=DEV) if M <= 128 else None
        h0, u0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        h1 = torch.empty_like(h0)
        gp = torch.empty(M, N, dtype=torch.float16, device=DEV)
        L().gemm(A, W, h0, epi=L().EPI_GELU, bias=bias, C2=u0, scratch=scratch)
        L().gemm(A, W, h1, epi=L().EPI_GELU_DG, bias=bias, C2=gp, scratch=scratch)
        assert torch.equal(h0, h1)
        ud = (A.double() @ W.double().t() + bias.double()).requires_grad_(True)
        torch.nn.functional.gelu(ud).sum().backward()
        close(gp, ud.grad, 2 ** -10, 2e-3 * math.sqrt(K / 64), f"gelu' saved as half ({M} x {N})")
        h2 = torch.empty_like(h0)
        L().gemm(A, W, h2, epi=L().EPI_GELU_DG, bias=bias, scratch=scratch)
        assert torch.equal(h0, h2)
        dY, Wt = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.05)
        dH = torch.empty_like(h0)
        L().gemm(dY, Wt, dH, epi=L().EPI_MULH, aux=gp, scratch=scratch)
        close(dH, (dY.double() @ Wt.double().t()) * gp.double(), 2 ** -8, 3e-3 * math.sqrt(K / 64), f"acc * saved gelu' ({M} x {N})")


@pytest.mark.parametrize("M,panels", [(12608, False), (2000, False), (1600, True)])
def test_fc2_dx_epilogue_riders(M, panels):
    """cara_gemm_args::er_*: the CARA_EPI_MULH launch also leaves, per row tile, the partial sums of dVs = dH^T T (+ column sums of
    dH) and dU = h^T G' -- dH itself bitwise what the launch without riders writes; the reduced products against fp64 on the bf16
    dH the kernel rounds to.  M = 2000: the last row tile is partial (at 160 rows its second wave row lies wholly outside)."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    N, K, Rp, rank = 3072, 768, 32, 16
    ldg = (M + 31) // 32 * 32
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    A2, B2 = rnd(M, Rp, seed=3, scale=0.3), rnd(N, Rp, seed=4, scale=0.3)
    gp = (torch.rand(M, N, generator=torch.Generator().manual_seed(5)) * 1.26 - 0.13).to(torch.float16).to(DEV)
    h = rnd(M, N, seed=8)
    hp = h.reshape(M, N // 32, 32).permute(1, 0, 2).contiguous() if panels else h
    Tt = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Gt = torch.zeros(Rp, ldg, dtype=torch.bfloat16, device=DEV)
    Tt[:rank, :M] = rnd(rank, M, seed=6)
    Gt[:rank, :M] = rnd(rank, M, seed=7)
    Tt[:, M:] = float("nan")    # (columns >= M are never to be read into a sum)
    Gt[:, M:] = float("nan")
    kw = dict(c_panels=M, ldc=N) if panels else {}
    shape = (N // 32, M, 32) if panels else (M, N)
    ref = torch.zeros(shape, dtype=torch.bfloat16, device=DEV)
    L().gemm(A, W, ref, epi=L().EPI_MULH, A2=A2, B2=B2, aux=gp, **kw)
    out = torch.zeros(shape, dtype=torch.bfloat16, device=DEV)
    out, sv, su, chunks = L().gemm(A, W, out, epi=L().EPI_MULH, A2=A2, B2=B2, aux=gp, epi_riders=(Tt, Gt, hp, True, M if panels else 0), **kw)
    rows = 160 if os.environ.get("CARA_ER_ROWS") == "160" else 128
    assert chunks == (M + rows - 1) // rows
    assert torch.equal(out, ref)
    Dv = torch.full((N, Rp), float("nan"), device=DEV)
    Du = torch.full((N, Rp), float("nan"), device=DEV)
    cs = torch.full((N,), float("nan"), device=DEV)
    tab = (L().TsReduce * 2)(L().TsReduce(p(sv), 0, p(Dv), p(cs), 1, M, N, Rp, 16, chunks),
                             L().TsReduce(p(su), 0, p(Du), None, 1, M, N, Rp, 16, chunks))
    L().check(lib.cara_tskinny_reduce_many(tab, 2, st()), "reduce many")
    torch.cuda.synchronize()
    dH = out.permute(1, 0, 2).reshape(M, N) if panels else out
    assert torch.count_nonzero(Dv[:, rank:]) == 0 and torch.count_nonzero(Du[:, rank:]) == 0
    close(Dv[:, :rank], dH.double().t() @ Tt[:rank, :M].double().t(), 1e-3, 2e-2, "dVs from the epilogue")
    close(cs, dH.double().sum(0), 1e-3, 2e-2, "dc from the epilogue")
    close(Du[:, :rank], h.double().t() @ Gt[:rank, :M].double().t(), 1e-3, 2e-2, "dU from the epilogue")
    # the same products by the kernels that re-read dH and h: equal to fp32 summation order
    Dv2 = torch.empty(N, Rp, device=DEV)
    cs2 = torch.empty(N, device=DEV)
    Tz = Tt.clone()
    Tz[:, M:] = 0
    L().tskinny_xtg(dH.contiguous(), Tz, Dv2, cs2, M=M)
    close(Dv[:, :rank], Dv2[:, :rank], 1e-5, 2e-3, "dVs: epilogue vs cara_tskinny_xtg")
    close(cs, cs2, 1e-5, 2e-3, "dc: epilogue vs cara_tskinny_xtg")
    # what cannot carry them says so
    small = torch.zeros(M, 768, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError):
        L().gemm(A, rnd(768, K, seed=8), small, epi=L().EPI_MULH, aux=gp[:, :768].contiguous(), epi_riders=(Tt, Gt, h[:, :768].contiguous(), True))


# ---- round 5: fp32 head, scaled cross-entropy, loss-scale state, AdamW skip word --------------------------------------------
@pytest.mark.parametrize("B,classes,D,tokens", [(64, 100, 768, 197), (2, 21843, 768, 197), (3, 10, 1024, 5), (1, 257, 256, 1)])
def test_head_forward_in_fp32(B, classes, D, tokens):
    """cara_head_forward: final LayerNorm of the cls rows (row stride tokens * D) + classifier head, fp32 throughout, against fp64
    torch; the 16-bit xn and the row statistics it leaves for the backward."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    x = rnd(B, tokens, D, seed=1, scale=2.0, dtype=torch.float32) + 0.3
    gamma, beta = rnd(D, seed=2, dtype=torch.float32) + 1.0, rnd(D, seed=3, scale=0.1, dtype=torch.float32)
    W, b = rnd(classes, D, seed=4, scale=0.02, dtype=torch.float32), rnd(classes, seed=5, scale=0.1, dtype=torch.float32)
    xn16 = torch.full((B, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    logits = torch.full((B, classes), float("nan"), device=DEV)
    L().check(lib.cara_head_forward(p(x), C.c_long(tokens * D), p(gamma), p(beta), p(W), p(b), p(xn16), p(mean), p(rstd), p(logits),
                                    B, classes, D, C.c_float(1e-6), st()), "head fwd")
    xd = x[:, 0].double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    xn = (xd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    close(logits, xn @ W.double().t() + b.double(), 2e-5, 2e-5, "head logits")
    close(mean, mu[:, 0], 1e-5, 1e-6, "mean")
    close(rstd, 1 / torch.sqrt(var[:, 0] + 1e-6), 1e-5, 1e-7, "rstd")
    close(xn16, xn, 2 ** -8, 1e-6, "xn16")


def test_cross_entropy_ex_amp_update_and_adamw_skip_word():
    """The three pieces of the device-side loss scaling: cara_cross_entropy_ex scales dlogits by dscale * *loss_scale and clears the
    found-inf word; cara_amp_update follows GradScaler's rule; an AdamW launch whose skip word is set changes nothing."""
    lib = L().lib()
    p, st = L().ptr, L().stream
    Bc, Cn = 16, 100
    logits = rnd(Bc, Cn, seed=5, scale=3.0, dtype=torch.float32)
    labels = torch.randint(0, Cn, (Bc,), generator=torch.Generator().manual_seed(6)).to(DEV)
    loss, dl = torch.empty(1 + Bc, device=DEV), torch.empty(Bc, Cn, device=DEV)
    amp = torch.tensor([512.0, 0.0, 0.0, 0.0], device=DEV)
    found = torch.ones(1, device=DEV)
    L().check(lib.cara_cross_entropy_ex(p(logits), p(labels), p(loss), p(dl), Bc, Cn, C.c_float(0.25), p(amp), p(found), st()), "xent ex")
    ld = logits.double().requires_grad_(True)
    rl = torch.nn.functional.cross_entropy(ld, labels)
    rl.backward()
    close(loss[0], rl.detach(), 1e-5, 1e-6, "loss stays unscaled")
    close(dl, ld.grad * 128.0, 1e-4, 1e-5, "dlogits x dscale x loss scale")
    assert found.item() == 0.0
    assert lib.cara_cross_entropy_ex(p(logits), p(labels), p(loss), p(dl), Bc, Cn, C.c_float(0.0), None, None, st()) != 0
    # GradScaler's rule: clean steps count up, the interval doubles the scale (capped), an overflow halves it and counts a skip
    args = (C.c_float(2.0), C.c_float(0.5), 3, C.c_float(2048.0))
    for want in ([512, 1, 0], [512, 2, 0], [1024, 0, 0]):
        L().check(lib.cara_amp_update(p(amp), p(found), *args, st()), "amp")
        assert amp[:3].tolist() == [float(v) for v in want], amp
    found.fill_(2.0)   # (the all-reduced word: a SUM over ranks)
    L().check(lib.cara_amp_update(p(amp), p(found), *args, st()), "amp")
    assert amp[:3].tolist() == [512.0, 0.0, 1.0]
    found.zero_()
    amp[0] = 2048.0
    for _ in range(3):
        L().check(lib.cara_amp_update(p(amp), p(found), *args, st()), "amp")
    assert amp[0].item() == 2048.0      # capped
    # AdamW: with the word set the launch leaves parameters and moments alone; cleared, it steps
    from cara_amd.optim import AdamW
    w = rnd(1500, seed=9, dtype=torch.float32).requires_grad_(True)
    w.grad = rnd(1500, seed=10, dtype=torch.float32)
    opt = AdamW([w], lr=1e-2)
    before = w.detach().clone()
    opt.skip_flag = found.fill_(1.0)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(w.detach(), before) and not opt.state[w]["exp_avg"].any()
    found.zero_()
    opt.step()
    assert not torch.equal(w.detach(), before)




def test_allreduce_flat_through_a_one_rank_rccl_communicator():
    """cara_allreduce_flat (SURVEY 8b's optional export): the step's one collective through RCCL for a host that owns an
    ncclComm_t.  One GPU here: a communicator of ONE rank, made with ctypes on librccl itself -- the SUM over one rank leaves the
    buffer as it was, on the caller's stream; NULL arguments are refused.  In a child process (tests/_rccl_one_rank.py): a
    communicator is process-wide state this test run has no other use for."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_one_rank.py")], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(here))
    if r.returncode == 77:
        pytest.skip(r.stdout.strip() or "no RCCL communicator in this environment")
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "allreduce ok" in r.stdout
