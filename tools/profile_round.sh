#!/bin/bash
# One round's evidence: the default bench line, the same command under rocprofv3 --kernel-trace --stats, and the
# idle-gap analysis of that trace.  Usage (on the GPU box): bash tools/profile_round.sh r01_h
set -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench_default_run.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err || exit 1
f=$(find gpurun_out/prof_${tag} -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats_final.csv
python3 tools/timeline.py gpurun_out/prof_${tag} --steps 10 --skip-last 3 > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/prof_${tag}
tail -c 400 gpurun_out/${tag}_bench_default_run.json; echo; head -5 gpurun_out/${tag}_timeline.txt
