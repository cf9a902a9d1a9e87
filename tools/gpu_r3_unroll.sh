# Matrix phase unrolled with immediate offsets (in-tree) against the build before it (build/prev), same box:
# the mono fused kernel alone, the stereo LM loop, and a short parity check.
set -e
OLD=$PWD/build/prev/libccalib.so
for rep in 1 2 3; do
  echo "mono new:"; timeout -k 10 100 python tools/kernel_bench.py 100000 200
  echo "mono old:"; CCALIB_LIB=$OLD timeout -k 10 100 python tools/kernel_bench.py 100000 200
done
echo "mono 11x8 new:"; BOARD=11x8 timeout -k 10 100 python tools/kernel_bench.py 20000 100
echo "mono 11x8 old:"; BOARD=11x8 CCALIB_LIB=$OLD timeout -k 10 100 python tools/kernel_bench.py 20000 100
for n in 10000 100000; do
  for rep in 1 2; do
    echo "pairs $n new:"; timeout -k 10 200 python tools/stereo_bench.py $n 20
    echo "pairs $n old:"; CCALIB_LIB=$OLD timeout -k 10 200 python tools/stereo_bench.py $n 20
  done
done
timeout -k 10 600 python -u -m pytest tests/test_gpu_stereo_blocks.py tests/test_golden.py tests/test_lm_trace.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t7.log 2>&1 || { tail -60 gpurun_out/t7.log; exit 1; }
tail -3 gpurun_out/t7.log
