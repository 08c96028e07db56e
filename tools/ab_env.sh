# same-box A/B of bench.py over values of one environment variable: tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo -n "$var=$v  "
  env $var=$v timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-info-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['config']['loss'])"
done; done
