#!/bin/bash
# round 5, run ZZ: the backward without the two memsets of the running-gradient buffers (last block on its cls rows) -- tests, same-box A/B (CARA_BWD_MEMSETS=1: with them)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== layernorm kernel tests + the whole model suite"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm" 2>&1 | tail -2 || exit 1
timeout -k 10 1100 python3 -m pytest tests/test_model_gpu.py -q -x 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== step A/B"
for round in 1 2 3; do
  for v in 1 0; do
    CARA_BWD_MEMSETS=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched --steps 40 > gpurun_out/r05_zz_${v}_${round}.json 2>> gpurun_out/r05_zz_err.txt || exit 1
    python3 - $v $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_zz_{m}_{r}.json").read().strip().split("\n")[-1])
print(f"memsets {m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f}), loss {d['config']['loss']:.6f}")
PY
  done
done
