set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu > gpurun_out/bench_prof.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_write.log 2>&1
