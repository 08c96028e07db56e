#!/bin/bash
# round 5, run U: the attention backward as four waves of 64 keys / queries (CARA_ATTN_BWD_V=3) -- tests, alone, in the step
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention tests with the four-wave backward"
CARA_ATTN_BWD_V=3 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -5 || exit 1
echo "== alone: seven waves (V=2) vs four waves (V=3)"
for i in 1 2 3; do
  for v in 2 3; do
    CARA_ATTN_BWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "bwd" | sed "s/^/V=$v  /"
  done
done
echo "== model tests (subset) with V=3"
CARA_ATTN_BWD_V=3 timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -x -k "depth2 or train_step_against or headline" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== step A/B"
for round in 1 2; do
  for v in 2 3; do
    CARA_ATTN_BWD_V=$v timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --no-precision-matched --steps 30 > gpurun_out/r05_u_${v}_${round}.json 2>> gpurun_out/r05_u_err.txt || exit 1
    python3 - $v $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_u_{m}_{r}.json").read().strip().split("\n")[-1])
s = {x["site"]: x["avg_launch_us"] for x in d["roofline_top"] + d["roofline_hbm"]}
print(f"V={m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f});  attn_bwd {s['attn_bwd']:.1f}")
PY
  done
done
