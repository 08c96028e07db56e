#!/bin/bash
# round 5, run K: the ring of three LDS slots (CARA_GEMM_RING3) -- bitwise test, then same-box A/B of the step per mask
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== ring test + the gemm tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "ring or gemm" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -4 || exit 1
echo "== step A/B"
for round in 1 2; do
  for mask in 0 1 2 3 4 7; do
    CARA_GEMM_RING3=$mask timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --no-precision-matched --steps 30 > gpurun_out/r05_k_${mask}_${round}.json 2>> gpurun_out/r05_k_err.txt || exit 1
    python3 - $mask $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_k_{m}_{r}.json").read().strip().split("\n")[-1])
s = {x["site"]: x["avg_launch_us"] for x in d["roofline_top"] + d["roofline_hbm"]}
print(f"mask {m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f}), fwd {d['config']['forward_only_ms']:.3f};  proj_fwd {s['proj_fwd']:.1f}  proj_bwd {s['proj_bwd']:.1f}  qkv_fwd {s['qkv_fwd']:.1f}  fc2_fwd {s['fc2_fwd']:.1f}")
PY
  done
done
