#!/bin/bash
# round 5, run M: the tail kernels after the second pass (few-row products with 4 / 8 waves, colred on all lanes) -- tests, timing of the stage-1 ranges, one step's kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== kernel tests + model subset"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "few_rows or factor or linear or gemm" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 || exit 1
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -x -k "depth2 or train_step or three_adamw or headline or cls or zero_init or graphed or other_orders" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
for split in 1 0; do
echo "== one step's kernels, CARA_GRAD_STAGE1_SPLIT=$split"
export CARA_GRAD_STAGE1_SPLIT=$split
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_m -- python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched \
  > gpurun_out/r05_m_bench_under_rocprof_$split.json 2> gpurun_out/r05_m_rocprof.err || exit 1
python3 tools/timeline.py gpurun_out/prof_r05_m --steps 10 --skip-last 3 --list > gpurun_out/r05_m_timeline_with_kernel_list_$split.txt 2>&1
rm -rf gpurun_out/prof_r05_m
grep "small_m_direct\|grad_stage\|reduce_many\|adamw\|prep_" gpurun_out/r05_m_timeline_with_kernel_list_$split.txt | cut -c1-120
done
