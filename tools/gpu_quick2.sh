set -e
timeout -k 10 120 python tools/kernel_bench.py 100000 200
CCALIB_NO_FUSED=1 timeout -k 10 120 python tools/kernel_bench.py 100000 200
timeout -k 10 200 python tools/stereo_bench.py 100000 20
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
