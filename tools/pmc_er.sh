#!/bin/bash
# counters of the fc2 dX launch with and without its epilogue riders (rocprofv3 --pmc passes over bench.py --steps 2, per kernel symbol):
# where the riders' 28 us go.  Usage (GPU box): bash tools/pmc_er.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_er
run() { tag=$1; n=$2; shift 2
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_er/${tag}_$n --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-info-legs > gpurun_out/pmc_er/${tag}_$n.log 2>&1 || echo "pass $tag $n failed"
}
for tag in on off; do
  if [ $tag = off ]; then export CARA_EPI_RIDERS=0; else export CARA_EPI_RIDERS=1; fi   # (the default is 0 since r04: the 'on' arm must say so)
  run $tag p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
  run $tag p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
  run $tag p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
  run $tag p4 FETCH_SIZE
  run $tag p5 WRITE_SIZE
  run $tag p6 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES
done
python3 - <<'PY'
import collections, csv, glob
for tag in ("on", "off"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"gpurun_out/pmc_er/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    print("=====", tag)
    for k, cs in sorted(acc.items()):
        if not ("gemm32_ts_kernel<6" in k or "gemm32_ts_kernel<4" in k or "gemm8_kernel" in k or "gemm32ft_ts_kernel<0, 1, true" in k or "tskinny_reduce" in k):
            continue
        n = max(v[1] for v in cs.values())
        print(k[:130], f"({n} samples)")
        for c in sorted(cs):
            print(f"      {c:34s} {cs[c][0] / cs[c][1]:16.0f}")
PY
rm -rf gpurun_out/pmc_er/*/*/*kernel_trace.csv gpurun_out/pmc_er/*/*/*agent_info.csv
