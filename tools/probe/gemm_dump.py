"""Outputs of cara_gemm_bf16 on the block's eight product shapes (seeded operands) saved to a file, so that two builds /
environment settings can be compared bit for bit: python tools/probe/gemm_dump.py out.pt ; python tools/probe/gemm_dump.py --cmp a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for k in a:
        eq = torch.equal(a[k], b[k])
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(f"{k:28s} {'bitwise equal' if eq else 'DIFFERENT'}  max|diff| {d:.3e}  ref max {a[k].float().abs().max().item():.3e}")
        bad += not eq
    sys.exit(1 if bad else 0)
from cara_amd import _lib as L
M0 = 64 * 197
SHAPES = [("qkv_fwd", 2304, 768, "bf16"), ("proj_fwd", 768, 768, "resid"), ("fc1_fwd", 3072, 768, "gelu"), ("fc2_fwd", 768, 3072, "resid"),
          ("fc2_bwd", 3072, 768, "dgelu"), ("fc1_bwd", 768, 3072, "bf16"), ("proj_bwd", 768, 768, "bf16"), ("qkv_bwd", 768, 2304, "bf16")]
dev = "cuda"; g = torch.Generator().manual_seed(0); out_all = {}
for M in (M0, 4 * 577):
    for name, N, K, epi in SHAPES:
        for apan in (False, True):
            A = torch.randn(M, K, generator=g).bfloat16().to(dev)
            B = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(dev)
            kw = dict(A2=torch.randn(M, 32, generator=g).bfloat16().to(dev), B2=(torch.randn(N, 32, generator=g) * 0.02).bfloat16().to(dev),
                      bias=torch.randn(N, generator=g).to(dev), Bp=L.pack_b_panels(B))
            Ain = A
            if apan:
                Ain = A.view(M, K // 32, 32).permute(1, 0, 2).contiguous(); kw.update(a_panels=M, M=M, K=K, lda=K)
            if epi == "bf16":
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); kw.update(epi=L.EPI_BF16)
                if apan: kw.update(c_panels=M)
            elif epi == "gelu":
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); kw.update(epi=L.EPI_GELU, C2=torch.zeros_like(out))
            elif epi == "dgelu":
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                kw.update(epi=L.EPI_DGELU, aux=torch.randn(M, N, generator=g).bfloat16().to(dev)); kw["bias"] = None
            else:
                out = torch.zeros(M, N, dtype=torch.float32, device=dev)
                kw.update(epi=L.EPI_RESID, aux=torch.randn(M, N, generator=g).to(dev), rowscale=torch.rand((M + 196) // 197, generator=g).to(dev), rows_per_sample=197)
            L.gemm(Ain, B, out, **kw)
            torch.cuda.synchronize()
            out_all[f"{name}_M{M}_{'pan' if apan else 'row'}"] = out.cpu()
            if epi == "gelu": out_all[f"{name}_M{M}_{'pan' if apan else 'row'}_u"] = kw["C2"].cpu()
torch.save(out_all, sys.argv[1])
print("saved", len(out_all), "outputs to", sys.argv[1])
