#!/bin/bash
# same-box A/B of the step: the 128 x 128 x 32 GEMM family everywhere (CARA_GEMM8=0) vs the 160 x 256 x 64 tile on the long-K narrow-N
# products, with and without helper waves; interleaved repetitions, per-site times of every variant.  Usage: tools/ab_g8.sh [reps] [steps]
REPS=${1:-3}; STEPS=${2:-40}
mkdir -p gpurun_out
for i in $(seq 1 $REPS); do
  for v in off on nohelp; do
    case $v in off) E="CARA_GEMM8=0";; on) E="CARA_GEMM8=160";; nohelp) E="CARA_GEMM8=160 CARA_GEMM8_HELPERS=0";; esac
    env $E timeout -k 10 300 python bench.py --steps $STEPS --warmup 5 --no-info-legs --no-cpu-baseline --all-sites > gpurun_out/abg8_${v}_$i.json 2>gpurun_out/abg8_err.txt || { echo "FAILED $v $i"; tail -5 gpurun_out/abg8_err.txt; exit 1; }
  done
done
python - <<'PY'
import json, glob, statistics
for v in ("off", "on", "nohelp"):
    runs = [json.loads(open(f).read().strip().split("\n")[-1]) for f in sorted(glob.glob(f"gpurun_out/abg8_{v}_*.json"))]
    ms = [r["ms_per_step"] for r in runs]
    fw = [r["config"]["forward_only_ms"] for r in runs]
    print(f"{v:7s} ms/step {' '.join('%.3f' % m for m in ms)}  median {statistics.median(ms):.3f}   forward-only {statistics.median(fw):.3f}  loss {runs[0]['config']['loss']}")
    sites = {}
    for r in runs:
        for t in r.get("roofline_top", []) + r.get("roofline_hbm", []):
            sites.setdefault(t["site"], []).append(t["avg_launch_us"])
    print("        " + "  ".join(f"{k} {statistics.median(x):.1f}" for k, x in sites.items()))
PY
