#!/bin/bash
# samples the GPU's clock and power (rocm-smi, read only) while bench.py runs: is the step power-limited?
python bench.py --steps 400 --warmup 5 --no-cpu-baseline > /tmp/bench_power.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)" | tr -s ' ' | head -12; echo ---; sleep 0.5; done
wait $pid
python -c "import json;d=json.loads(open('/tmp/bench_power.json').readline());print('ms/step',d['ms_per_step'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
