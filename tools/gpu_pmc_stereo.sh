set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { d=$1; shift; rm -rf gpurun_out/$d; timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/$d -o p -- python3 tools/stereo_bench.py 100000 4 > gpurun_out/$d.log 2>&1; }
run spmcA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
run spmcB SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM
run spmcC TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr
python3 tools/rocpd_summary.py pmc gpurun_out/spmcA/p_results.db gpurun_out/spmcB/p_results.db gpurun_out/spmcC/p_results.db gpurun_out/pmc_stereo.json | grep -i "stereo\|err_wave\|compose"
rm -rf gpurun_out/spmcA gpurun_out/spmcB gpurun_out/spmcC
