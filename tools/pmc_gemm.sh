export TMPDIR=/tmp
run() { # name tile counters...
  name=$1; tile=$2; shift 2
  CARA_GEMM_TILE=$tile timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$name --output-format csv -- python3 tools/gemm_bench.py --iters 2 > gpurun_out/pmc_$name.log 2>&1 || echo "pass $name failed"
}
run a256 256 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES
run b256 256 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run c256 256 SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
run b128 128 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE
ls gpurun_out/pmc_*/*/ | head -30
