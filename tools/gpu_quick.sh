set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_init.log 2>&1 || { tail -60 gpurun_out/t_init.log; exit 1; }
tail -3 gpurun_out/t_init.log
timeout -k 10 200 python tools/wholecall_bench.py 100000
