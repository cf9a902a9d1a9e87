# The retry's back-substitution taken along in the first trial's kernel (stereo), against a launch of its own, same box.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_stereo_blocks.py tests/test_golden.py tests/test_gpu_stereo_full.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t6.log 2>&1 || { tail -60 gpurun_out/t6.log; exit 1; }
tail -3 gpurun_out/t6.log
for n in 10000 100000; do
  for rep in 1 2; do
    echo "pairs $n dual:"; timeout -k 10 200 python tools/stereo_bench.py $n 20
    echo "pairs $n separate:"; CCALIB_NO_DUAL_BACKSUB=1 timeout -k 10 200 python tools/stereo_bench.py $n 20
  done
done
