#!/bin/bash
# round 5, final evidence: the whole GPU suite, the default bench line + rocprofv3 stats + timeline (tools/profile_round.sh), every site, HBM traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== smoke"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== the whole GPU suite"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -s 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = self\|^$" > gpurun_out/r05_fin_gpu_tests.txt; tail -4 gpurun_out/r05_fin_gpu_tests.txt
if grep -q "Memory access fault" gpurun_out/r05_fin_gpu_tests.txt; then echo FAULT; exit 1; fi
echo "== profile_round"
bash tools/profile_round.sh r05_fin 2>&1 | tail -8
echo "== all sites"
timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 > gpurun_out/r05_fin_all_sites.json 2>> gpurun_out/r05_fin_err.txt
echo "== traffic"
CARA_PMC_TAG=r05_fin bash tools/pmc_traffic.sh 2>&1 | tail -18
cp gpurun_out/pmc_traffic_summary.txt gpurun_out/r05_fin_pmc_traffic_summary.txt
echo "== ViT-L/16 @384, batch 32"
timeout -k 10 400 python3 bench.py --model vit_large_patch16_384 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/r05_fin_vitl_bench.json 2>> gpurun_out/r05_fin_err.txt; tail -c 300 gpurun_out/r05_fin_vitl_bench.json
