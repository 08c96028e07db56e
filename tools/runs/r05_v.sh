#!/bin/bash
# round 5, run V: the four-wave attention backward, two vector chains per fence -- alone
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CARA_ATTN_BWD_V=3 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2 || exit 1
for i in 1 2 3; do
  for v in 2 3; do
    CARA_ATTN_BWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "bwd" | sed "s/^/V=$v  /"
  done
done
