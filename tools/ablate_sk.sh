#!/bin/bash
# stream-K GEMM ablations (timing only; results are wrong for ablate != 0): see SkPlan::ablate in gemm_sk.hip
for ab in ${ABLATES:-0 32 34 36 40 38 46}; do
  echo "== CARA_GEMM_ABLATE=$ab"
  CARA_GEMM_ABLATE=$ab timeout -k 10 120 python tools/gemm_bench.py --iters 20 --sk --shape 8192,8192,8192 --shape 4096,4096,768 --shape 12608,3072,768,gelu --shape 12608,768,3072 2>&1 | grep -v amdgpu.ids
done
