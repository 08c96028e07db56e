#!/bin/bash
# r05 counter evidence: MFMA-busy and instruction mix of the attention kernels and of the top GEMM symbols, from
# rocprofv3 --pmc passes with --kernel-trace only (separate passes per counter group) over a SHORT run of the real step
# (bench.py --steps 2): per kernel symbol, averages over its launches.  Usage (GPU box): bash tools/pmc_r05.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_r05
run() { n=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_r05/$n --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-info-legs --no-precision-matched $PMC_EXTRA > gpurun_out/pmc_r05/$n.log 2>&1 || echo "pass $n failed"
}
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_r05/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
keys = ["attn_fwd_persist", "attn_fwd_p2", "attn_fwd_long", "attn_bwd_dkv", "attn_bwd_dq", "attn_bwd_fused", "gemm8_kernel", "gemm8_ts_kernel", "gemm32_kernel<2", "gemm32_ts_kernel<4", "gemm32ft_ts_kernel<0, 1, true", "gemm32_kernel<0", "gemm32ft_kernel<3", "ln_fwd_kernelILi3ELb1", "ln_bwd_kernelILi3ELb1"]
with open("gpurun_out/r05_pmc_attn_and_gemm.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc (three separate passes) over bench.py --steps 2: per kernel symbol, counter averages per launch\n")
    out.write("# MFMA-busy share = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (per-SE sums: both scale alike); VALU : MFMA = SQ_INSTS_VALU / SQ_INSTS_MFMA\n")
    for k, cs in sorted(acc.items()):
        if not any(x in k for x in keys):
            continue
        g = {c: v[0] / v[1] for c, v in cs.items()}
        n = max(v[1] for v in cs.values())
        out.write(k[:150] + f"   ({n} launches)\n")
        if g.get("SQ_BUSY_CYCLES"):
            out.write(f"   MFMA-busy share of busy cycles: {g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / g['SQ_BUSY_CYCLES']:.3f}\n")
        if g.get("SQ_INSTS_MFMA"):
            out.write(f"   VALU : MFMA instructions = {g.get('SQ_INSTS_VALU', 0) / g['SQ_INSTS_MFMA']:.2f} : 1;  LDS : MFMA = {g.get('SQ_INSTS_LDS', 0) / g['SQ_INSTS_MFMA']:.2f} : 1\n")
        if g.get("SQ_LDS_IDX_ACTIVE"):
            out.write(f"   LDS bank-conflict cycles / LDS active cycles: {g.get('SQ_LDS_BANK_CONFLICT', 0) / g['SQ_LDS_IDX_ACTIVE']:.3f}\n")
        for c in sorted(g):
            out.write(f"      {c:34s} {g[c]:16.0f}\n")
print(open("gpurun_out/r05_pmc_attn_and_gemm.txt").read()[:6000])
PY
rm -rf gpurun_out/pmc_r05/*/*/*kernel_trace.csv
