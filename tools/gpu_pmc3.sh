set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/$d -o p -- python3 tools/kernel_bench.py 100000 3 > gpurun_out/$d.log 2>&1; }
run pmcA SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
run pmcB SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
run pmcC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU
python3 tools/rocpd_summary.py pmc gpurun_out/pmcA/p_results.db gpurun_out/pmcB/p_results.db gpurun_out/pmcC/p_results.db gpurun_out/pmc_new.json > /dev/null
