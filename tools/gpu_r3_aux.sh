# The doomed trial on a second stream beside its retry (default) against one stream (CCALIB_AUX_STREAM=0), same box.
set -e
for n in 10000 100000; do
  for rep in 1 2 3; do
    echo "pairs $n aux:"; timeout -k 10 200 python tools/stereo_bench.py $n 20
    echo "pairs $n one stream:"; CCALIB_AUX_STREAM=0 timeout -k 10 200 python tools/stereo_bench.py $n 20
  done
done
for rep in 1 2; do
  echo "mono whole call aux:"; timeout -k 10 200 python tools/wholecall_bench.py 100000
  echo "mono whole call one stream:"; CCALIB_AUX_STREAM=0 timeout -k 10 200 python tools/wholecall_bench.py 100000
done
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t12.log 2>&1 || { tail -60 gpurun_out/t12.log; exit 1; }
tail -3 gpurun_out/t12.log
