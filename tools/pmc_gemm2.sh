#!/bin/bash
# rocprofv3 PMC passes over the default GEMM kernels (instruction mix and wait reasons)
export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_g2_$name --output-format csv -- python3 tools/gemm_bench.py --iters 3 > gpurun_out/pmc_g2_$name.log 2>&1 || echo "pass $name failed"
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA
run c SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
