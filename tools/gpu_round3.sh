# End-of-round measurement set: GPU tests, the bench line, rocprofv3 kernel statistics of the bench and of the stereo
# loop (10k and 100k pairs, with the idle gaps between kernels), HBM byte counters for both, SQ counters of the mono
# Jacobian kernel for this build and for the build of commit a4f74dc (build/r2/libccalib.so), the whole-call timings.
# Each --pmc pass is a run of its own.  Results under gpurun_out/; tools/save_profiles.sh r03 copies the summaries.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-300 gpurun_out/bench.json
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu > gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_summary.py kernels gpurun_out/prof/bench_results.db gpurun_out/bench_stats.csv > gpurun_out/bench_stats.txt
for n in 100000 10000; do
  rm -rf gpurun_out/prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py $n 20 > gpurun_out/stereo_prof_$n.log 2>&1
  python3 tools/rocpd_summary.py kernels gpurun_out/prof/st_results.db gpurun_out/stereo_stats_$n.csv > gpurun_out/stereo_stats_$n.txt
  python3 tools/rocpd_gaps.py gpurun_out/prof/st_results.db 20 > gpurun_out/stereo_gaps_$n.txt 2>&1 || true
done
rm -rf gpurun_out/prof
pmc() { d=$1; shift; rm -rf gpurun_out/$d; timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/$d -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$d.log 2>&1; }
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
python3 tools/rocpd_summary.py pmc gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db gpurun_out/pmc_sq/p_results.db gpurun_out/bench_pmc.json > /dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
# stereo HBM bytes per launch, keyed by the number of pairs (bench.py reads this shape)
for n in 100000 10000; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/sp_$c
    timeout -s KILL 120 rocprofv3 --pmc $c -d gpurun_out/sp_$c -o p -- python3 tools/stereo_bench.py $n 4 > gpurun_out/sp_$c.log 2>&1
  done
  python3 tools/rocpd_summary.py pmc gpurun_out/sp_FETCH_SIZE/p_results.db gpurun_out/sp_WRITE_SIZE/p_results.db gpurun_out/pmc_stereo_$n.json > /dev/null
  rm -rf gpurun_out/sp_FETCH_SIZE gpurun_out/sp_WRITE_SIZE
done
python3 - <<'PY'
import json
json.dump({n: json.load(open("gpurun_out/pmc_stereo_%s.json" % n)) for n in ("10000", "100000")}, open("gpurun_out/pmc_stereo.json", "w"), indent=1)
PY
# the mono Jacobian kernel alone: where its SIMD time goes, this build and the one of commit a4f74dc
sq() { tag=$1; lib=$2; rm -rf gpurun_out/sq_$tag; if [ -n "$lib" ]; then export CCALIB_LIB=$lib CCALIB_OLD_LIB_OK=1; else unset CCALIB_LIB CCALIB_OLD_LIB_OK; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/sq_$tag -o p -- python3 tools/kernel_bench.py 100000 20 > gpurun_out/sq_$tag.log 2>&1
  python3 tools/rocpd_summary.py pmc gpurun_out/sq_$tag/p_results.db gpurun_out/sq_$tag.json > /dev/null; rm -rf gpurun_out/sq_$tag; unset CCALIB_LIB CCALIB_OLD_LIB_OK; }
sq r03 ""
sq r02 $PWD/build/r2/libccalib.so
echo "kernel alone, this build / commit a4f74dc:"; timeout -k 10 100 python tools/kernel_bench.py 100000 200; CCALIB_OLD_LIB_OK=1 CCALIB_LIB=$PWD/build/r2/libccalib.so timeout -k 10 100 python tools/kernel_bench.py 100000 200
CCALIB_TIMING=1 timeout -k 10 200 python tools/wholecall_bench.py 100000 > gpurun_out/wholecall.json 2> gpurun_out/wholecall_phases.txt
timeout -k 10 200 python tools/wholecall_bench.py stereo 10000 >> gpurun_out/wholecall.json
cat gpurun_out/wholecall.json
head -12 gpurun_out/bench_stats.txt
head -14 gpurun_out/stereo_stats_100000.txt
