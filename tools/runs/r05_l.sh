#!/bin/bash
# round 5, run L: the few-row products in one launch (CARA_SMALL_M_DIRECT) and the gradient scatter's shorter chains -- tests, A/B, one step's kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== kernel tests (few rows, factors) + model subset"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "few_rows or factor or linear or gemm" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 || exit 1
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -x -k "depth2 or train_step or three_adamw or headline or cls or zero_init or graphed" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== step A/B: CARA_SMALL_M_DIRECT"
for round in 1 2 3; do
  for v in 0 1; do
    CARA_SMALL_M_DIRECT=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched --steps 40 > gpurun_out/r05_l_${v}_${round}.json 2>> gpurun_out/r05_l_err.txt || exit 1
    python3 - $v $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_l_{m}_{r}.json").read().strip().split("\n")[-1])
print(f"direct {m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f}), fwd {d['config']['forward_only_ms']:.3f}")
PY
  done
done
echo "== one step's kernels"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_l -- python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched \
  > gpurun_out/r05_l_bench_under_rocprof.json 2> gpurun_out/r05_l_rocprof.err || exit 1
python3 tools/timeline.py gpurun_out/prof_r05_l --steps 10 --skip-last 3 --list > gpurun_out/r05_l_timeline_with_kernel_list.txt 2>&1
cp "$(find gpurun_out/prof_r05_l -name '*kernel_stats.csv' | head -1)" gpurun_out/r05_l_kernel_stats.csv
rm -rf gpurun_out/prof_r05_l
head -3 gpurun_out/r05_l_timeline_with_kernel_list.txt
