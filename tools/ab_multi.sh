#!/bin/bash
# Same-box A/B of bench.py under several environment settings, interleaved and repeated:
#   bash tools/ab_multi.sh "<bench flags>" <reps> "VAR=a VAR2=b" "VAR=c" ...
# prints ms_per_step (mean), median, images/s per setting and repetition.
flags=$1; reps=$2; shift 2
for r in $(seq 1 $reps); do
  for envs in "$@"; do
    out=$( ( export $envs; timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-legs $flags 2>/dev/null ) | tail -1 )
    echo "$envs | rep $r | $(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('ms_per_step_median'), d['value'], 'fwd', d['config']['forward_only_ms'], 'loss', round(d['config']['loss'],5))")"
  done
done
