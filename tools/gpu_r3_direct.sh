# Sums published by every column-sum workgroup itself ({sum, number} pairs in pinned memory, default) against the flag
# paths (CCALIB_DIRECT_SUMS=0), same box: mono loop at three sizes, stereo loop at two; then the complete GPU suite.
set -e
for v in 100000 10000 1000; do
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --no-cpu --views $v | python3 tools/bench_line.py
    CCALIB_DIRECT_SUMS=0 timeout -k 10 300 python3 bench.py --no-cpu --views $v | python3 tools/bench_line.py
  done
done
for n in 10000 100000; do
  for rep in 1 2; do
    timeout -k 10 200 python tools/stereo_bench.py $n 20 | cut -c40-120
    CCALIB_DIRECT_SUMS=0 timeout -k 10 200 python tools/stereo_bench.py $n 20 | cut -c40-120
  done
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/t17.log 2>&1 || { tail -60 gpurun_out/t17.log; exit 1; }
tail -3 gpurun_out/t17.log
