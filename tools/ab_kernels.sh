#!/bin/bash
# Per-kernel average durations of the bench under two environments, side by side.
# Usage (GPU box): bash tools/ab_kernels.sh "ENV_A=.." "ENV_B=.."
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for envs in "$1" "$2"; do
  i=$((i+1))
  ( export $envs; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$i -- python3 bench.py --no-cpu-baseline --no-info-legs --steps 10 --warmup 3 > gpurun_out/abk_$i.log 2>&1 ) || exit 1
  cp "$(find gpurun_out/abk_$i -name '*kernel_stats.csv' | head -1)" gpurun_out/abk_$i.csv
  rm -rf gpurun_out/abk_$i
done
python3 - <<'PY'
import csv
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3)
    return d
a, b = load("gpurun_out/abk_1.csv"), load("gpurun_out/abk_2.csv")
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0, 0))[1] + b.get(n, (0, 0, 0))[1]))
print(f"{'kernel':70s} {'calls':>6s} {'A avg us':>9s} {'B avg us':>9s} {'A tot ms':>9s} {'B tot ms':>9s}")
for n in names[:22]:
    ca, ta, va = a.get(n, (0, 0, 0)); cb, tb, vb = b.get(n, (0, 0, 0))
    print(f"{n[:70]:70s} {ca:6d} {va:9.1f} {vb:9.1f} {ta:9.2f} {tb:9.2f}")
print("sum of kernel time: A %.1f ms, B %.1f ms" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
PY
