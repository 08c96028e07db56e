#!/bin/bash
# Build libcara_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcara_hip.so
SRCS="lib.hip gemm.hip gemm8.hip skinny.hip norm_misc.hip attention.hip factors.hip dropout_exact.hip dense_delta.hip optim.hip"
[ -f vit.hip ] && SRCS="$SRCS vit.hip"
OBJS=""
mkdir -p build
pids=()
for s in $SRCS; do
  o=build/${s%.hip}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ gemm_epilogue.h -nt "$o" ] || [ tskinny_body.h -nt "$o" ] || [ gemm8.h -nt "$o" ] || [ ../../include/cara_hip.h -nt "$o" ]; then
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $(realpath $OUT)"
