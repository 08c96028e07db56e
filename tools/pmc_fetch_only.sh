export TMPDIR=/tmp
for gmv in 1 8; do
CARA_GEMM_GROUPM=$gmv timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_g$gmv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-info-legs > gpurun_out/pmc_fetch_g$gmv.log 2>&1 || echo pass failed
done
