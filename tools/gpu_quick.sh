set -e
timeout -k 10 120 python tools/kernel_bench.py 100000 200
CCALIB_GRID_FUSED=512 timeout -k 10 120 python tools/kernel_bench.py 100000 200
CCALIB_TIMING=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['rms_px'], d['roofline']['launch_ms'])"
