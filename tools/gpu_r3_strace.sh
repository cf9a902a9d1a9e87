# Kernel statistics and idle gaps of the stereo LM loop at 100k and 10k pairs (rocprofv3 kernel trace).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 100000 10000; do
  rm -rf gpurun_out/strace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/strace -o t -- python3 tools/stereo_bench.py $n 20 > gpurun_out/strace_$n.log 2>&1
  tail -1 gpurun_out/strace_$n.log
  python3 tools/rocpd_summary.py kernels gpurun_out/strace/t_results.db gpurun_out/stereo_stats_$n.csv > gpurun_out/stereo_stats_$n.txt
  python3 tools/rocpd_gaps.py gpurun_out/strace/t_results.db > gpurun_out/stereo_gaps_$n.txt 2>&1 || true
  head -16 gpurun_out/stereo_stats_$n.txt
done
rm -rf gpurun_out/strace
