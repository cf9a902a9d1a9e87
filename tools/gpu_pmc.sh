set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d gpurun_out/pmc1 -o p -- python3 tools/kernel_bench.py 100000 3 > gpurun_out/pmc1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM -d gpurun_out/pmc2 -o p -- python3 tools/kernel_bench.py 100000 3 > gpurun_out/pmc2.log 2>&1
ls gpurun_out/pmc1 gpurun_out/pmc2
