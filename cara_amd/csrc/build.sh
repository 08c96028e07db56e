#!/bin/bash
# Build libcara_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   build.sh        -> ../libcara_hip.so      (bf16 operands: the product)
#   build.sh f16    -> ../libcara_hip_f16.so  (the same sources with IEEE-half operands, -DCARA_F16_OPERANDS: precision = "fp16")
#   build.sh all    -> both
set -euo pipefail
cd "$(dirname "$0")"
SRCS="lib.hip gemm.hip gemm8.hip skinny.hip norm_misc.hip attention.hip factors.hip dropout_exact.hip dense_delta.hip optim.hip linear.hip"
[ -f vit.hip ] && SRCS="$SRCS vit.hip"
build_one() {   # $1 = object dir, $2 = output, $3.. = extra flags
  local dir=$1 out=$2; shift 2
  local OBJS="" pids=()
  mkdir -p "$dir"
  for s in $SRCS; do
    local o=$dir/${s%.hip}.o
    OBJS="$OBJS $o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ gemm_epilogue.h -nt "$o" ] || [ tskinny_body.h -nt "$o" ] || [ gemm8.h -nt "$o" ] || [ ../../include/cara_hip.h -nt "$o" ]; then
      # attention.hip: no packed fp32 arithmetic beside the MFMAs (scalar source, no SLP vectorisation: a v_pk_* there costs ~22
      # cycles more than the two scalar instructions it replaces, MI355X_MICROARCH.md; r05: backward -1 us in the step)
      local extra=""
      [ "$s" = attention.hip ] && extra="-fno-slp-vectorize"
      hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $extra "$@" -c "$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $OBJS
  echo "built $(realpath $out)"
}
case "${1:-bf16}" in
  bf16) build_one build ../libcara_hip.so ;;
  f16) build_one build_f16 ../libcara_hip_f16.so -DCARA_F16_OPERANDS ;;
  all) build_one build ../libcara_hip.so; build_one build_f16 ../libcara_hip_f16.so -DCARA_F16_OPERANDS ;;
  *) echo "usage: build.sh [bf16|f16|all]"; exit 2 ;;
esac
