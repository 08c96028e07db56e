#!/bin/bash
# round 5, box E: graph mode (test + bench), the rehearsal bench with its precision_matched object, counters of the attention kernels
# (ViT-B step and ViT-L/16 @384: the long-sequence kernels' first profile)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
fault() { if grep -q "Memory access fault" gpurun_out/r05_e_log.txt; then echo FAULT; exit 1; fi; }
echo "== tests"
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "graph or adamw or rehearsal or amp or head_forward or three_adamw or overflow" 2>&1 | tail -6 || exit 1
fault
echo "== bench, eager then --graph (same box)"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-info-legs --steps 40 > gpurun_out/r05_e_bench_eager.json 2> gpurun_out/r05_e_err.txt || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-info-legs --steps 40 --graph > gpurun_out/r05_e_bench_graph.json 2>> gpurun_out/r05_e_err.txt || exit 1
python3 - <<'PY'
import json
for n in ("eager", "graph"):
    d = json.load(open(f"gpurun_out/r05_e_bench_{n}.json"))
    pm = d.get("precision_matched", {})
    print(n, d["value"], d["ms_per_step"], d["ms_per_step_median"], "fwd", d["config"]["forward_only_ms"], d["config"].get("launch"), "| fp16:", pm.get("value"), pm.get("ms_per_step"), pm.get("vs_bf16_step"), pm.get("error"))
    print("   roofline", d["roofline"]["kernel"][:60], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
fault
echo "== ViT-L/16 @384 bs 32: kernel trace (the long-sequence attention kernels)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vitl -- python3 bench.py --model vit_large_patch16_384 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/r05_e_vitl_bench.json 2>> gpurun_out/r05_e_err.txt || echo "vitl profile failed"
f=$(find gpurun_out/prof_vitl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_e_vitl_kernel_stats.csv && head -14 "$f" | cut -c1-200
rm -rf gpurun_out/prof_vitl
fault
echo "== counters: ViT-B step"
bash tools/pmc_r05.sh > gpurun_out/r05_e_pmc.log 2>&1; tail -5 gpurun_out/r05_e_pmc.log | cut -c1-200
mv gpurun_out/r05_pmc_attn_and_gemm.txt gpurun_out/r05_pmc_vitb.txt 2>/dev/null
echo "== counters: ViT-L step"
PMC_EXTRA="--model vit_large_patch16_384 --batch 32" bash tools/pmc_r05.sh > gpurun_out/r05_e_pmc_l.log 2>&1; tail -3 gpurun_out/r05_e_pmc_l.log | cut -c1-200
mv gpurun_out/r05_pmc_attn_and_gemm.txt gpurun_out/r05_pmc_vitl.txt 2>/dev/null
rm -rf gpurun_out/pmc_r05
fault
