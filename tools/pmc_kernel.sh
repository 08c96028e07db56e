#!/bin/bash
# rocprofv3 PMC passes (one counter group per run: no tracing domains besides --kernel-trace) over ONE GEMM shape of
# tools/gemm_bench.py (or PROG=tools/attn_bench.py KFILTER=attn).  Usage (GPU box): bash tools/pmc_kernel.sh <tag> <args...>; summary: gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
run() { n=$1; shift
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$tag/$n --output-format csv -- python3 ${PROG:-tools/gemm_bench.py} --iters 3 $ARGS > gpurun_out/pmc_$tag/$n.log 2>&1 || echo "pass $n failed"
}
ARGS="$*"
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
run p4 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run p5 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run p6 TD_TD_BUSY_sum TD_TC_STALL_sum
run p7 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run p8 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run p9 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run p10 TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_CYCLE_sum
run p11 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
python3 - "$tag" <<'PY'
import collections, csv, glob, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        import os
        if os.environ.get("KFILTER", "gemm32") not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(f"gpurun_out/pmc_{tag}.txt", "w") as out:
    for k, cs in acc.items():
        out.write(k[:110] + "\n")
        for c in sorted(cs):
            v, n = cs[c]
            out.write(f"  {c:40s} {v / n:16.0f}   ({n} launches)\n")
print(open(f"gpurun_out/pmc_{tag}.txt").read())
PY
