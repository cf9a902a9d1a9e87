set -e
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['rms_px'], d['roofline']['launch_ms'], d['roofline']['traffic'])"
timeout -k 10 200 python tools/stereo_bench.py 100000 20
