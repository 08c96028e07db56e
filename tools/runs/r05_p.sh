#!/bin/bash
# round 5, run P: where the 207-us gap of run O's profiled bench comes from (brackets on every 6th vs every 12th block), then the round's evidence
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for every in 12 6; do
  echo "== brackets on every ${every}th block"
  export CARA_BENCH_PROFILE_EVERY=$every
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/r05_p_plain_$every.json 2>> gpurun_out/r05_p_err.txt || exit 1
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_p -- python3 bench.py --no-cpu-baseline --no-info-legs --no-precision-matched \
    > gpurun_out/r05_p_rocprof_$every.json 2> gpurun_out/r05_p_rocprof.err || exit 1
  python3 tools/timeline.py gpurun_out/prof_r05_p --steps 10 --skip-last 3 --list > gpurun_out/r05_p_timeline_$every.txt 2>&1
  rm -rf gpurun_out/prof_r05_p
  head -6 gpurun_out/r05_p_timeline_$every.txt | cut -c1-170
  python3 - $every <<'PY'
import json, sys
e = sys.argv[1]
for k in ("plain", "rocprof"):
    d = json.loads(open(f"gpurun_out/r05_p_{k}_{e}.json").read().strip().split("\n")[-1])
    print(f"  {k}: mean {d['ms_per_step']:.3f} median {d['ms_per_step_median']:.3f} ms")
PY
done
unset CARA_BENCH_PROFILE_EVERY
