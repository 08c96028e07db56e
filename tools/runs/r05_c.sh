#!/bin/bash
# round 5, third box: the pipelined attention kernels (tests, A/B, stamps), the two-stream step, the per-site fp16 / bf16 comparison
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== attention tests"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention" 2>&1 | tail -5 || exit 1
echo "== attention micro-benchmark: round-3 bodies (V=1) vs the specialised schedules (V=2)"
for v in 1 2 1 2; do CARA_ATTN_FWD_V=$v CARA_ATTN_BWD_V=$v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep "attention" | sed "s/^/V=$v /"; done
echo "== stamps, V=2"
CARA_LIB_PATH=tools/probe/libcara_attnstamps.so timeout -k 10 120 python3 tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids
echo "== stamps, V=1"
CARA_ATTN_FWD_V=1 CARA_ATTN_BWD_V=1 CARA_LIB_PATH=tools/probe/libcara_attnstamps.so timeout -k 10 120 python3 tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids
echo "== model tests (subset)"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -q -m gpu -s -k "headline or train_step or three_adamw or overflow or depth2 or fp16_precision or graph" 2>&1 | grep -v Warning | grep -E "rel-L2|passed|failed|FAILED|Error|error|assert" | tail -40
echo "== two streams"
timeout -k 10 400 python3 tools/two_stream_step.py 2>&1 | grep -v amdgpu.ids | tail -20
echo "== per-site bf16 / fp16"
timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 > gpurun_out/r05_c_sites_bf16.json 2> gpurun_out/r05_c_err.txt
timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 --precision fp16 > gpurun_out/r05_c_sites_fp16.json 2>> gpurun_out/r05_c_err.txt
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/r05_c_sites_bf16.json")); b = json.load(open("gpurun_out/r05_c_sites_fp16.json"))
print("bf16", a["ms_per_step"], a["config"]["forward_only_ms"], "fp16", b["ms_per_step"], b["config"]["forward_only_ms"])
sa = {e["site"]: e for e in a["roofline_top"] + a["roofline_hbm"]}; sb = {e["site"]: e for e in b["roofline_top"] + b["roofline_hbm"]}
tot = 0
for n in sa:
    d = (sb[n]["avg_launch_us"] - sa[n]["avg_launch_us"]) * sa[n]["launches_per_step"]
    tot += d
    print(f"   {n:10s} bf16 {sa[n]['avg_launch_us']:7.2f} fp16 {sb[n]['avg_launch_us']:7.2f} us x {sa[n]['launches_per_step']:2d} -> {d:+7.1f} us/step")
print("   sum of site deltas", round(tot, 1), "us/step")
PY
