set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_init.py -m gpu -q --timeout 200 --timeout-method thread > gpurun_out/t_init.log 2>&1 || { tail -60 gpurun_out/t_init.log; exit 1; }
tail -3 gpurun_out/t_init.log
timeout -k 10 200 python tools/wholecall_bench.py 100000
