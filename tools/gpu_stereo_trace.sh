set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/strace
CCALIB_TIMING=1 timeout -k 10 200 python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_plain.log 2>&1
tail -5 gpurun_out/stereo_plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/strace -o t -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/strace.log 2>&1
tail -2 gpurun_out/strace.log
python3 tools/rocpd_summary.py kernels gpurun_out/strace/t_results.db gpurun_out/stereo_stats.csv > /dev/null
rm -rf gpurun_out/strace
