#!/bin/bash
# round 5, run R: the long-sequence attention kernels (577 tokens, ViT-L/16 @384) with key tiles in pairs / loops unrolled twice
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 || exit 1
echo "== long attention alone (B 32, N 577, H 16): previous attention.hip vs this one"
for i in 1 2 3; do
  CARA_LIB_PATH=tools/probe/libcara_prev_attn.so timeout -k 10 120 python3 tools/attn_bench.py --shape 32,577,16 --iters 20 2>&1 | grep "attention" | sed 's/^/prev  /'
  timeout -k 10 120 python3 tools/attn_bench.py --shape 32,577,16 --iters 20 2>&1 | grep "attention" | sed 's/^/this  /'
done
echo "== ViT-L tests"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -m gpu -s -k "vit_large" 2>&1 | grep -E "rel-L2|passed|failed" | tail -8
echo "== ViT-L step"
for v in prev this; do
  if [ $v = prev ]; then export CARA_LIB_PATH=tools/probe/libcara_prev_attn.so; else unset CARA_LIB_PATH; fi
  timeout -k 10 400 python3 bench.py --model vit_large_patch16_384 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/r05_r_vitl_$v.json 2>> gpurun_out/r05_r_err.txt || exit 1
  python3 -c "
import json
d = json.loads(open('gpurun_out/r05_r_vitl_$v.json').read().strip().split('\n')[-1])
print('$v', d['ms_per_step'], 'ms per step,', d['value'], 'images/s')"
done
