# End-of-round measurement set: GPU tests, the bench line, rocprofv3 kernel statistics for the bench and the stereo
# loop, HBM byte counters and wait-share counters (each --pmc pass in a run of its own).  Results under gpurun_out/.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
timeout -k 10 500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-400 gpurun_out/bench.json
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu > gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/bench_results.db gpurun_out/bench_stats.csv > gpurun_out/bench_stats.txt
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo_stats.csv > gpurun_out/stereo_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof/st_results.db > gpurun_out/stereo_gaps.txt 2>&1 || true
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 10000 20 > gpurun_out/stereo10k_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo10k_stats.csv > gpurun_out/stereo10k_stats.txt
rm -rf gpurun_out/prof
pmc() { d=$1; shift; rm -rf gpurun_out/$d; timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/$d -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$d.log 2>&1; }
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
python3 tools/rocpd_summary.py pmc gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db gpurun_out/pmc_sq/p_results.db gpurun_out/bench_pmc.json > /dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
head -12 gpurun_out/bench_stats.txt
head -12 gpurun_out/stereo_stats.txt
