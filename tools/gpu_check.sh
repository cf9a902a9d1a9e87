# quick check after a kernel change: dominant kernel alone, the bench line, the GPU test suite
set -e
timeout -k 10 120 python tools/kernel_bench.py 100000 200
timeout -k 10 200 python bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['rms_px'])"
timeout -k 10 200 python tools/stereo_bench.py 100000 20
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
