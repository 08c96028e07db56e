#!/bin/bash
# round 5, second box: the whole GPU suite on the ABI-14 tree, the forward-attention A/B, the per-site fp16 / bf16 comparison
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention micro-benchmark: round-3 body (V=1) vs specialised schedule (V=2)"
for v in 1 2 1 2; do CARA_ATTN_FWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention fwd" | sed "s/^/V=$v /"; done
echo "== kernel tests"
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_gemm8_gpu.py -q -m gpu -x 2>&1 | tail -15 || exit 1
echo "== model tests"
timeout -k 10 1100 python3 -m pytest tests/test_model_gpu.py tests/test_exact_dropout.py tests/test_cara_api.py tests/test_data.py tests/test_checkpoint.py -q -m gpu -s 2>&1 | grep -v Warning | grep -E "rel-L2|rel |passed|failed|FAILED|Error|error|assert|differ|loss " | tail -120
