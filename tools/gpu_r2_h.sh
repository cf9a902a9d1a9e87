set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_bygrid.py gpurun_out/prof/st_results.db colsum
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bn -- python3 bench.py --no-cpu --steps 40 > gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_bygrid.py gpurun_out/prof/bn_results.db colsum
python3 tools/rocpd_summary.py kernels gpurun_out/prof/bn_results.db gpurun_out/bench_stats.csv | head -8
rm -rf gpurun_out/prof
