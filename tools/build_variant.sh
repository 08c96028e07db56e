#!/bin/bash
# A/B build of the library with extra compiler flags: tools/build_variant.sh NAME -DFLAG=... -> tools/probe/libcara_NAME.so
# (same-box comparison: bash tools/ab_env.sh CARA_LIB_PATH cara_amd/libcara_hip.so tools/probe/libcara_NAME.so)
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../cara_amd/csrc"
mkdir -p build_$name
OBJS=""
pids=()
for s in lib gemm gemm8 skinny norm_misc attention factors dropout_exact dense_delta optim linear vit; do
  o=build_$name/$s.o
  OBJS="$OBJS $o"
  extra=""; [ $s = attention ] && extra="-fno-slp-vectorize"   # (as cara_amd/csrc/build.sh)
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $extra "$@" -c $s.hip -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
mkdir -p ../../tools/probe
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe/libcara_$name.so $OBJS
echo "built tools/probe/libcara_$name.so"
