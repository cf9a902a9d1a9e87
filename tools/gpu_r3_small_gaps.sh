# Kernel trace with idle gaps of a whole monocular call at a small size (launch- and turn-around-bound).
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o sm -- python3 tools/wholecall_bench.py ${1:-20} > gpurun_out/small_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/sm_results.db gpurun_out/small_stats.csv > gpurun_out/small_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof/sm_results.db 24 > gpurun_out/small_gaps.txt 2>&1 || true
head -16 gpurun_out/small_stats.txt
head -30 gpurun_out/small_gaps.txt
