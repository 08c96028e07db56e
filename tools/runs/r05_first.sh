#!/bin/bash
# round 5, first box: same-box per-site A/B of the two operand builds + attention micro-benchmark
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 > gpurun_out/r05_a_sites_bf16_$i.json 2> gpurun_out/r05_a_err.txt || exit 1
  timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 --precision fp16 > gpurun_out/r05_a_sites_fp16_$i.json 2>> gpurun_out/r05_a_err.txt || exit 1
done
timeout -k 10 120 python3 tools/attn_bench.py > gpurun_out/r05_a_attn_bench.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for i in (1, 2):
    a = json.load(open(f"gpurun_out/r05_a_sites_bf16_{i}.json")); b = json.load(open(f"gpurun_out/r05_a_sites_fp16_{i}.json"))
    print(i, "bf16", a["ms_per_step"], a["config"]["forward_only_ms"], "fp16", b["ms_per_step"], b["config"]["forward_only_ms"])
    sa = {e["site"]: e for e in a["roofline_top"] + a["roofline_hbm"]}; sb = {e["site"]: e for e in b["roofline_top"] + b["roofline_hbm"]}
    tot = 0
    for n in sa:
        d = (sb[n]["avg_launch_us"] - sa[n]["avg_launch_us"]) * sa[n]["launches_per_step"]
        tot += d
        print(f"   {n:10s} bf16 {sa[n]['avg_launch_us']:7.2f} fp16 {sb[n]['avg_launch_us']:7.2f} us x {sa[n]['launches_per_step']:2d} -> {d:+7.1f} us/step")
    print("   sum of site deltas", round(tot, 1), "us/step")
PY
cat gpurun_out/r05_a_attn_bench.txt
