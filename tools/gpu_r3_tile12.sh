# The 12-column fused kernel (all cameras in one launch) against the 16-column kernel, same box: parity tests first,
# then the stereo LM loop at 10k and 100k pairs each way.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_stereo_blocks.py tests/test_golden.py tests/test_gpu_stereo_full.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t5.log 2>&1 || { tail -60 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
for n in 10000 100000; do
  for rep in 1 2; do
    echo "pairs $n tile12:"; timeout -k 10 200 python tools/stereo_bench.py $n 20
    echo "pairs $n tile16:"; CCALIB_NO_TILE12=1 timeout -k 10 200 python tools/stereo_bench.py $n 20
  done
done
