set -e
timeout -k 10 300 python bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['valu_only_path'])"
