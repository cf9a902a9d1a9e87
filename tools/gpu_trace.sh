set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/trace -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/trace.log 2>&1
ls gpurun_out/trace
