set -e
export TMPDIR=/tmp
for n in 10000 100000; do
  rm -rf gpurun_out/strace
  timeout -k 10 200 python3 tools/stereo_bench.py $n 20
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/strace -o t -- python3 tools/stereo_bench.py $n 20 > gpurun_out/strace.log 2>&1
  python3 tools/rocpd_summary.py kernels gpurun_out/strace/t_results.db gpurun_out/stereo_stats_$n.csv > /dev/null
  python3 tools/rocpd_gaps.py gpurun_out/strace/t_results.db 20 > gpurun_out/stereo_gaps_$n.txt
  cp gpurun_out/strace/t_results.db gpurun_out/strace_$n.db
done
rm -rf gpurun_out/strace
