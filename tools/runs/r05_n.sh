#!/bin/bash
# round 5, run N: packed fp32 vector arithmetic in the attention kernels (v_pk_fma / v_pk_mul / v_pk_add on aligned pairs), the
# gradient scatter's stage 1 on layer groups -- tests, attention alone and the step against the previous attention.hip (same box)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== kernel tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 || exit 1
echo "== attention alone: previous attention.hip vs this one"
for i in 1 2 3; do
  CARA_LIB_PATH=tools/probe/libcara_prev_attn.so timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention" | sed 's/^/prev  /'
  timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention" | sed 's/^/this  /'
done
echo "== model tests (subset)"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -x -k "depth2 or train_step or three_adamw or headline or cls or zero_init or graphed or other_orders or odd" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== step A/B"
for round in 1 2 3; do
  for v in prev this; do
    if [ $v = prev ]; then export CARA_LIB_PATH=tools/probe/libcara_prev_attn.so; else unset CARA_LIB_PATH; fi
    timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --no-precision-matched --steps 30 > gpurun_out/r05_n_${v}_${round}.json 2>> gpurun_out/r05_n_err.txt || exit 1
    python3 - $v $round <<'PY'
import json, sys
m, r = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r05_n_{m}_{r}.json").read().strip().split("\n")[-1])
s = {x["site"]: x["avg_launch_us"] for x in d["roofline_top"] + d["roofline_hbm"]}
print(f"{m} round {r}: {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median']:.3f}), fwd {d['config']['forward_only_ms']:.3f};  attn_fwd {s['attn_fwd']:.1f}  attn_bwd {s['attn_bwd']:.1f}")
PY
  done
done
unset CARA_LIB_PATH
echo "== tail kernels"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_n -- python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched \
  > gpurun_out/r05_n_bench_under_rocprof.json 2> gpurun_out/r05_n_rocprof.err || exit 1
python3 tools/timeline.py gpurun_out/prof_r05_n --steps 10 --skip-last 3 --list > gpurun_out/r05_n_timeline_with_kernel_list.txt 2>&1
rm -rf gpurun_out/prof_r05_n
grep "small_m_direct\|grad_stage\|reduce_many\|adamw\|prep_" gpurun_out/r05_n_timeline_with_kernel_list.txt | cut -c1-120
