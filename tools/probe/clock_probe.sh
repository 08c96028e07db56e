#!/bin/bash
# average sclk / power (rocm-smi samples) while bench.py runs under a value of one environment variable:
#   tools/probe/clock_probe.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 500 --warmup 5 --no-cpu-baseline > /tmp/bench_clk.json 2>/dev/null &
  pid=$!
  sleep 7
  : > /tmp/clk.txt
  for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" >> /tmp/clk.txt; sleep 0.15; done
  wait $pid
  python3 - "$var=$v" <<'P'
import re,sys,json
t=open('/tmp/clk.txt').read()
s=[int(x) for x in re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)',t)]
p=[float(x) for x in re.findall(r'Power \(W\): ([\d.]+)',t)]
d=json.loads(open('/tmp/bench_clk.json').readline())
print(sys.argv[1], 'ms/step', d['ms_per_step'], 'sclk MHz avg', round(sum(s)/max(len(s),1),1), 'min', min(s), 'max', max(s), 'power W avg', round(sum(p)/max(len(p),1),1))
P
done
