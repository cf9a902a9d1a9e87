# no-alloc entry point test, then the whole GPU suite, then the bench line
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/na.log 2>&1 || { tail -40 gpurun_out/na.log; exit 1; }
tail -3 gpurun_out/na.log
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -14 gpurun_out/t.log
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
