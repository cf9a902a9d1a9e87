# bench line + stereo kernel trace (baseline before the round-2 kernel work)
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/bench.json'));print(d['ms_per_step'],d['value']);print(json.dumps(d['stereo'],indent=1));print(json.dumps(d['cpu_baseline']['dense_one_core'],indent=1))"
rm -rf gpurun_out/strace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/strace -o t -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/strace.log 2>&1
tail -2 gpurun_out/strace.log
python3 tools/rocpd_summary.py kernels gpurun_out/strace/t_results.db gpurun_out/stereo_stats.csv
rm -rf gpurun_out/strace
