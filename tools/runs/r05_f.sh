#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== graph test"
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -q -m gpu -x -k "graphed" 2>&1 | grep -v Warning | tail -40
echo "== attention tests + bench (forward with the deferred epilogue)"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_gemm8_gpu.py -q -m gpu -x 2>&1 | tail -4
for v in 1 2 1 2; do CARA_ATTN_FWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention fwd" | sed "s/^/V=$v /"; done
if grep -q "Memory access fault" gpurun_out/r05_f_log.txt; then exit 1; fi
