#!/bin/bash
# Ordered kernel list of one training step (rocprofv3 --kernel-trace of a short bench run): bash tools/run_list.sh TAG
set -o pipefail
tag=${1:-list}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-info-legs \
  > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_rocprof.err || exit 1
python3 tools/timeline.py gpurun_out/prof_${tag} --steps 4 --skip-last 3 --list > gpurun_out/${tag}_steplist.txt 2>&1
rm -rf gpurun_out/prof_${tag}
head -3 gpurun_out/${tag}_steplist.txt
