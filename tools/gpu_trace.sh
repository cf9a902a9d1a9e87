set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/trace -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/trace.log 2>&1
grep -a '"metric"' gpurun_out/trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
