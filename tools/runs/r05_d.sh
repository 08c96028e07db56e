#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== two streams"
timeout -k 10 500 python3 tools/two_stream_step.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|_run_forward" | tail -20
if grep -q "Memory access fault" gpurun_out/r05_d_log.txt 2>/dev/null; then exit 1; fi
