cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --no-cpu-baseline --no-info-legs --steps 8 --warmup 3 > gpurun_out/tl.log 2>&1 && python3 tools/timeline.py gpurun_out/tl --steps 5 > gpurun_out/tl.txt 2>&1; tail -1 gpurun_out/tl.log | cut -c1-200; cat gpurun_out/tl.txt; rm -rf gpurun_out/tl
