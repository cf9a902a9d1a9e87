# Kernel trace with idle gaps of the mono LM loop at a small size (1 000 views: launch- and turn-around-bound).
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o sm -- python3 bench.py --no-cpu --views ${1:-1000} > gpurun_out/small_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/sm_results.db gpurun_out/small_stats.csv > gpurun_out/small_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof/sm_results.db 20 > gpurun_out/small_gaps.txt 2>&1 || true
head -20 gpurun_out/small_stats.txt
head -60 gpurun_out/small_gaps.txt
CCALIB_TIMING=1 timeout -k 10 200 python3 bench.py --no-cpu --views ${1:-1000} 2>&1 | grep -a "ccalib\]" | tail -5
