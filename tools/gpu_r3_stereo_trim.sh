# Stereo LM loop after the launch trims (two dampings' sums side by side, both cameras' residual pass in one launch,
# the composition folded into the back-substitution) against the build before them, same box; then parity tests.
set -e
OLD=$PWD/build/prev/libccalib.so
for n in 10000 100000; do
  for rep in 1 2 3; do
    echo "pairs $n new:"; timeout -k 10 200 python tools/stereo_bench.py $n 20
    echo "pairs $n old:"; CCALIB_LIB=$OLD timeout -k 10 200 python tools/stereo_bench.py $n 20
  done
  echo "pairs $n new, separate compose:"; CCALIB_FOLD_COMPOSE=0 timeout -k 10 200 python tools/stereo_bench.py $n 20
done
timeout -k 10 900 python -u -m pytest tests/test_gpu_stereo_blocks.py tests/test_gpu_stereo_full.py tests/test_golden.py tests/test_lm_trace.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t11.log 2>&1 || { tail -60 gpurun_out/t11.log; exit 1; }
tail -3 gpurun_out/t11.log
