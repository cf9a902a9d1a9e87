set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stereo_full.py -m gpu -x -q --timeout 300 --timeout-method thread -k "stereo or Stereo" > gpurun_out/t_stereo.log 2>&1 || { tail -30 gpurun_out/t_stereo.log; exit 1; }
tail -2 gpurun_out/t_stereo.log
timeout -k 10 200 python tools/stereo_bench.py 100000 20

timeout -k 10 200 python tools/stereo_bench.py 10000 20
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo_stats.csv | head -12
rm -rf gpurun_out/prof
timeout -k 10 300 python bench.py --no-cpu --steps 40 > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err
python3 -c "import json;d=json.loads(open('gpurun_out/bench_g.json').read().strip().splitlines()[-1]);print('mono ms_per_step',d['ms_per_step'],'stereo',{k:v for k,v in d.get('stereo',{}).items() if 'ms' in k or 'pairs' in k})"
