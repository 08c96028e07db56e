#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention + head tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention or head_forward" 2>&1 | tail -4 || exit 1
echo "== backward: four barriers per head (V=1) vs two (V=2)"
for v in 1 2 1 2 1 2; do CARA_ATTN_BWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention bwd" | sed "s/^/V=$v /"; done
echo "== stamps, V=2"
CARA_LIB_PATH=tools/probe/libcara_attnstamps.so timeout -k 10 120 python3 tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids | sed -n '1,30p'
echo "== model tests (subset) + repeated attention tests (a race would come and go)"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -m gpu -k "headline or train_step_against or depth2 or fp16_precision_at or vit_large_384_against" 2>&1 | tail -3
for i in 1 2 3; do timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention_fwd_bwd" 2>&1 | tail -1; done
echo "== step A/B"
for v in 1 2 1 2; do CARA_ATTN_BWD_V=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('V=$v', d['ms_per_step'], d['ms_per_step_median'])"; done
if grep -q "Memory access fault" gpurun_out/r05_i_log.txt; then exit 1; fi
