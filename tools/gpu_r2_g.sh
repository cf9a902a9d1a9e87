set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stereo_full.py -m gpu -x -q --timeout 300 --timeout-method thread -k "stereo or Stereo" > gpurun_out/t_stereo.log 2>&1 || { tail -30 gpurun_out/t_stereo.log; exit 1; }
tail -2 gpurun_out/t_stereo.log
timeout -k 10 200 python tools/stereo_bench.py 100000 20
CCALIB_SUMS_OLD=1 timeout -k 10 200 python tools/stereo_bench.py 100000 20
timeout -k 10 200 python tools/stereo_bench.py 10000 20
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo_stats.csv | head -12
rm -rf gpurun_out/prof
