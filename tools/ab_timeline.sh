#!/bin/bash
# Idle-gap analysis (tools/timeline.py) of the bench under two environments.
# Usage (GPU box): bash tools/ab_timeline.sh "ENV_A=.." "ENV_B=.."
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for envs in "$1" "$2" "$1" "$2"; do
  i=$((i+1))
  ( export $envs; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abt_$i -- python3 bench.py --no-cpu-baseline --no-info-legs --steps 10 --warmup 3 > gpurun_out/abt_$i.log 2>&1 ) || exit 1
  echo "== $envs: $(grep -h '^{"metric"' gpurun_out/abt_$i.log | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, forward only", d["config"]["forward_only_ms"])')"
  python3 tools/timeline.py gpurun_out/abt_$i --steps 8 > gpurun_out/abt_$i.txt 2>&1; sed -n 1,8p gpurun_out/abt_$i.txt
  rm -rf gpurun_out/abt_$i
done
