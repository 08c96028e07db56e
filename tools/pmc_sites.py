#!/usr/bin/env python3
"""HBM-side bytes per launch of the bracketed kernel sites, from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and
WRITE_SIZE need separate passes: TCC slots).  gfx950 correction of MI355X_MICROARCH.md (section HBM): FETCH_SIZE counts
half the bytes of wide coalesced reads -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The i-th launch of a kernel name
is the same launch in both passes.  Usage: pmc_sites.py <fetch dir> <write dir> <out.json>"""
import collections, csv, glob, json, os, sys


def series(path, counter):
    f = sorted(glob.glob(path + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # (the newest run's file)
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f[-1])):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024)
    return out


fetch, write = series(sys.argv[1], "FETCH_SIZE"), series(sys.argv[2], "WRITE_SIZE")
MB = 1e6
# site -> (substring of the kernel name, predicate on (read bytes x2, written bytes) that picks the full-size training launches)
SITES = {
    "fc1_fwd": ("gemm32_kernel<2,", lambda r, w: w > 120 * MB),                 # two bf16 outputs of [12608, 3072]
    "qkv_fwd": ("gemm32_kernel<0,", lambda r, w: w > 50 * MB),
    # (r04: fc2 forward and qkv dX run on the 160 x 256 x 64 tile of gemm8.hip; the gemm32ft patterns cover CARA_GEMM8=0 runs)
    "fc2_fwd": [("gemm8_kernel<", lambda r, w: r > 100 * MB), ("gemm32ft_kernel<3", lambda r, w: r > 100 * MB)],   # reads h (77 MB) + the fp32 residual
    "proj_fwd": ("gemm32ft_kernel<3", lambda r, w: 30 * MB < r <= 100 * MB),
    "fc2_bwd": ("gemm32_ts_kernel<4, true", lambda r, w: w > 50 * MB),
    # (with G' inside -- the default since round 3 -- fc1 / qkv dX run gemm32ft_ts_kernel<BF16, NU = 1, COLSUM, ...>: COLSUM tells them apart)
    "fc1_bwd": [("gemm32ft_ts_kernel<0, 1, true", lambda r, w: w > 10 * MB),
                ("gemm32_ts_kernel<0, true", lambda r, w: r > 150 * MB)],       # dH as GEMM operand and as the products' operand
    "proj_bwd": ("gemm32_ts_kernel<0, true", lambda r, w: 20 * MB < r <= 150 * MB),
    "qkv_bwd": [("gemm8_ts_kernel<", lambda r, w: w > 10 * MB), ("gemm32ft_ts_kernel<0, 1, false", lambda r, w: w > 10 * MB),
                ("gemm32_ts_kernel<0, false", lambda r, w: w > 10 * MB)],
    "attn_fwd": [("attn_fwd_p2_kernel", lambda r, w: True), ("attn_fwd_persist_kernel", lambda r, w: True)],
    "attn_bwd": ("attn_bwd_fused_kernel", lambda r, w: True),
    "ln_fwd": ("ln_fwd_kernelILi3ELb1", lambda r, w: w > 15 * MB),
    "ln_bwd": ("ln_bwd_kernelILi3ELb1", lambda r, w: w > 40 * MB),
    "skinny_bwd": ("skinny_xu_sliced_kernel", lambda r, w: r > 40 * MB),
}
out = {}
for site, pats in SITES.items():
    rows = []
    for key, pick in (pats if isinstance(pats, list) else [pats]):
        for name in fetch:
            if key in name and name in write:
                for f, w in zip(fetch[name], write[name]):
                    if pick(2 * f, w):
                        rows.append((2 * f, w))
    if rows:
        n = len(rows)
        r, w = sum(x for x, _ in rows) / n, sum(y for _, y in rows) / n
        out[site] = {"launches": n, "read_mb_corrected": round(r / MB, 1), "write_mb": round(w / MB, 1),
                     "hbm_bytes_per_launch_corrected": int(r + w)}
meta = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py --steps 2 (tools/pmc_traffic.sh); "
                   "read = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads)"}
# which kernels this was measured on (bench.py compares it with the sources of the run that quotes the number)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import kernel_source_sha16
    meta["_meta"] = {"kernel_source_sha16": kernel_source_sha16(), "profile": os.environ.get("CARA_PMC_TAG", "unnamed")}
except Exception as e:   # noqa: BLE001
    meta["_meta"] = {"kernel_source_sha16": None, "profile": f"hash unavailable: {e}"}
meta.update(out)
json.dump(meta, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:14s} launches {v['launches']:4d}  read {v['read_mb_corrected']:8.1f} MB  write {v['write_mb']:8.1f} MB")
