#!/bin/bash
# round 5, final evidence: the whole GPU suite, the default bench line + rocprofv3 stats + timeline (tools/profile_round.sh), every site, HBM traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== smoke"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = " | tail -3 || exit 1
echo "== the whole GPU suite"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -s 2>&1 | grep -v "Warning\|amdgpu.ids\|logits = self\|^$" > gpurun_out/r05_o_gpu_tests.txt; tail -4 gpurun_out/r05_o_gpu_tests.txt
if grep -q "Memory access fault" gpurun_out/r05_o_gpu_tests.txt; then echo FAULT; exit 1; fi
echo "== profile_round"
bash tools/profile_round.sh r05_o 2>&1 | tail -8
echo "== all sites"
timeout -k 10 300 python3 bench.py --all-sites --no-cpu-baseline --no-info-legs --steps 30 > gpurun_out/r05_o_all_sites.json 2>> gpurun_out/r05_o_err.txt
echo "== traffic"
CARA_PMC_TAG=r05_o bash tools/pmc_traffic.sh 2>&1 | tail -18
cp gpurun_out/pmc_traffic_summary.txt gpurun_out/r05_o_pmc_traffic_summary.txt
