set -e
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
for b in 9x6 11x8 17x12; do echo board=$b; BOARD=$b timeout -k 10 120 python tools/kernel_bench.py 50000 50; BOARD=$b CCALIB_NO_FUSED=1 timeout -k 10 120 python tools/kernel_bench.py 50000 50; done
