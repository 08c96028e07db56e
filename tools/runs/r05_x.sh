#!/bin/bash
# round 5, run X: no packed fp32 arithmetic in the attention kernels (scalar source, -fno-slp-vectorize) vs the packed build -- alone and in the step
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2 || exit 1
echo "== alone: packed (prev) vs scalar (this); backward V=2 (seven waves) and V=3 (four waves)"
for i in 1 2 3; do
  CARA_LIB_PATH=tools/probe/libcara_prev_attn.so timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention" | sed 's/^/prev      /'
  timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention" | sed 's/^/this      /'
  CARA_ATTN_BWD_V=3 timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "bwd" | sed 's/^/this V=3  /'
done
echo "== step A/B"
for round in 1 2 3; do
  for v in prev this; do
    if [ $v = prev ]; then export CARA_LIB_PATH=tools/probe/libcara_prev_attn.so; else unset CARA_LIB_PATH; fi
    timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --no-precision-matched --steps 30 > gpurun_out/r05_x_${v}_${round}.json 2>> gpurun_out/r05_x_err.txt || exit 1
    python3 - $v $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_x_{m}_{r}.json").read().strip().split("\n")[-1])
s = {x["site"]: x["avg_launch_us"] for x in d["roofline_top"] + d["roofline_hbm"]}
print(f"{m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f});  attn_fwd {s['attn_fwd']:.1f}  attn_bwd {s['attn_bwd']:.1f}")
PY
  done
done
