cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# HBM traffic of the bracketed kernel sites (bench.py roofline.traffic reads profiles/pmc_traffic_by_site.json):
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), kernel-trace only.
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/pmc_fetch.log 2>&1 || echo fetch pass failed
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-info-legs --no-precision-matched > gpurun_out/pmc_write.log 2>&1 || echo write pass failed
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/ | head
python3 tools/pmc_sites.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic_by_site.json > gpurun_out/pmc_traffic_summary.txt 2>&1; cat gpurun_out/pmc_traffic_summary.txt
rm -rf gpurun_out/pmc_fetch/*/*kernel_trace.csv gpurun_out/pmc_write/*/*kernel_trace.csv
