#!/bin/bash
# same-box A/B of the step over a list of environment settings ("NAME=VAL NAME2=VAL2" per variant, "-" = defaults), interleaved
# repetitions, medians and per-site times.  Usage: tools/ab_env_list.sh REPS STEPS "-" "CARA_X=1" "CARA_X=2 CARA_Y=0" ...
REPS=$1; STEPS=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 $REPS); do
  n=0
  for v in "$@"; do
    n=$((n+1))
    if [ "$v" = "-" ]; then E=""; else E="$v"; fi
    env $E timeout -k 10 300 python bench.py --steps $STEPS --warmup 5 --no-info-legs --no-cpu-baseline --all-sites > gpurun_out/abl_${n}_$i.json 2>gpurun_out/abl_err.txt || { echo "FAILED variant $n rep $i"; tail -5 gpurun_out/abl_err.txt; exit 1; }
  done
done
python - "$@" <<'PY'
import json, glob, statistics, sys
for n, v in enumerate(sys.argv[1:], 1):
    runs = [json.loads(open(f).read().strip().split("\n")[-1]) for f in sorted(glob.glob(f"gpurun_out/abl_{n}_*.json"))]
    ms = [r["ms_per_step"] for r in runs]
    fw = [r["config"]["forward_only_ms"] for r in runs]
    print(f"[{v}] ms/step {' '.join('%.3f' % m for m in ms)}  median {statistics.median(ms):.3f}   forward-only {statistics.median(fw):.3f}  loss {runs[0]['config']['loss']:.6f}")
    sites = {}
    for r in runs:
        for t in r.get("roofline_top", []) + r.get("roofline_hbm", []):
            sites.setdefault(t["site"], []).append(t["avg_launch_us"])
    print("        " + "  ".join(f"{k} {statistics.median(x):.1f}" for k, x in sites.items()))
PY
