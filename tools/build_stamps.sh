#!/bin/bash
# Diagnostic build of the library with in-kernel time stamps in the GEMM (tools/gemm_stamps.py): tools/probe/libcara_stamps.so
set -euo pipefail
cd "$(dirname "$0")/../cara_amd/csrc"
mkdir -p build_stamps
OBJS=""
pids=()
for s in lib gemm skinny norm_misc attention factors dropout_exact dense_delta optim vit; do
  o=build_stamps/$s.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s.hip" -nt "$o" ] || [ gemm_epilogue.h -nt "$o" ] || [ common.h -nt "$o" ] || [ tskinny_body.h -nt "$o" ]; then
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DCARA_GEMM_STAMPS -c $s.hip -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
mkdir -p ../../tools/probe
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe/libcara_stamps.so $OBJS
echo "built tools/probe/libcara_stamps.so"
