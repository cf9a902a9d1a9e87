set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python tools/stereo_bench.py 100000 20
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo_stats.csv > gpurun_out/stereo_stats.txt
grep -E "err_wave|backsub|view_sums" gpurun_out/stereo_stats.txt
rm -rf gpurun_out/prof
