set -e
timeout -k 10 120 python tools/kernel_bench.py 100000 20
timeout -k 10 120 python tools/kernel_bench.py 100000 500
timeout -k 10 120 python tools/kernel_bench.py 100000 3000
CCALIB_NO_FUSED=1 timeout -k 10 120 python tools/kernel_bench.py 100000 1000
rocm-smi --showclocks 2>/dev/null | head -20 || true
