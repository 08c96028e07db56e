# per-site launch times of the step for values of one environment variable: tools/site_times.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-info-legs --all-sites 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('ms/step',d['ms_per_step'])
for t in d['roofline_top']+d['roofline_hbm']: print(f\"{t['site']:12s} {t['avg_launch_us']:8.2f} us x {t['launches_per_step']}\")
"
done
