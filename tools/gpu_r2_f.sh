set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
CCALIB_NO_CHAIN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu --steps 40 > gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/bench_results.db gpurun_out/bench_stats.csv | head -8
rm -rf gpurun_out/prof
