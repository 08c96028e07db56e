#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention" 2>&1 | tail -4 || exit 1
echo "== long-sequence micro-benchmark (B 32, N 577, H 16)"
timeout -k 10 200 python3 tools/attn_bench.py --shape 32,577,16 2>&1 | grep attention
echo "== ViT-L model tests"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -m gpu -s -k "vit_large" 2>&1 | grep -E "rel-L2|passed|failed" | tail -8
echo "== ViT-L bench"
timeout -k 10 400 python3 bench.py --model vit_large_patch16_384 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/r05_h_vitl_bench.json 2> gpurun_out/r05_h_err.txt; python3 -c "
import json; d=json.load(open('gpurun_out/r05_h_vitl_bench.json')); print('ViT-L', d['value'], d['ms_per_step'])"
echo "== traffic (kernel sources final)"
CARA_PMC_TAG=r05_h bash tools/pmc_traffic.sh 2>&1 | tail -15
if grep -rq "Memory access fault" gpurun_out/r05_h_log.txt; then exit 1; fi
