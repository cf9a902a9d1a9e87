set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "running_ahead or config1 or full_size" > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/bench.json 2> gpurun_out/bench.err
python3 -c "import json;d=json.load(open('gpurun_out/bench.json'));print('chain   ',d['ms_per_step'],d['value'],d['roofline']['launch_ms'],d['jac_passes'],d['err_passes'])"
CCALIB_NO_CHAIN=1 timeout -k 10 300 python bench.py --no-cpu > gpurun_out/bench_nochain.json 2>> gpurun_out/bench.err
python3 -c "import json;d=json.load(open('gpurun_out/bench_nochain.json'));print('no chain',d['ms_per_step'],d['value'],d['roofline']['launch_ms'])"
done
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu --steps 40 > gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/bench_results.db gpurun_out/bench_stats.csv | head -12
python3 tools/rocpd_gaps.py gpurun_out/prof/bench_results.db 30
rm -rf gpurun_out/prof
