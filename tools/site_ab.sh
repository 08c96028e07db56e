#!/bin/bash
# per-site launch times of the step under several environment settings (same box): bash tools/site_ab.sh "<bench flags>" "VAR=a" "VAR=b" ...
flags=$1; shift
for envs in "$@"; do
  echo "== $envs"
  ( export $envs; timeout -k 10 200 python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-info-legs --all-sites $flags 2>/dev/null ) | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(' step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'fwd', d['config']['forward_only_ms'])
for t in d['roofline_top']+d['roofline_hbm']: print('   %-11s %7.2f us x%3d' % (t['site'], t['avg_launch_us'], t['launches_per_step']))"
done
