set -e
export TMPDIR=/tmp
# the mono Jacobian kernel alone: where its SIMD time goes, this build and round 2's
sq() { tag=$1; lib=$2; rm -rf gpurun_out/sq_$tag; if [ -n "$lib" ]; then export CCALIB_LIB=$lib CCALIB_OLD_LIB_OK=1; else unset CCALIB_LIB CCALIB_OLD_LIB_OK; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/sq_$tag -o p -- python3 tools/kernel_bench.py 100000 20 > gpurun_out/sq_$tag.log 2>&1
  python3 tools/rocpd_summary.py pmc gpurun_out/sq_$tag/p_results.db gpurun_out/sq_$tag.json > /dev/null; rm -rf gpurun_out/sq_$tag; unset CCALIB_LIB CCALIB_OLD_LIB_OK; }
sq r03 ""
sq r02 $PWD/build/r2/libccalib.so
echo "kernel alone, this build / round 2:"; timeout -k 10 100 python tools/kernel_bench.py 100000 200; CCALIB_OLD_LIB_OK=1 CCALIB_LIB=$PWD/build/r2/libccalib.so timeout -k 10 100 python tools/kernel_bench.py 100000 200
CCALIB_TIMING=1 timeout -k 10 200 python tools/wholecall_bench.py 100000 > gpurun_out/wholecall.json 2> gpurun_out/wholecall_phases.txt
timeout -k 10 200 python tools/wholecall_bench.py stereo 10000 >> gpurun_out/wholecall.json
cat gpurun_out/wholecall.json
head -12 gpurun_out/bench_stats.txt
head -14 gpurun_out/stereo_stats_100000.txt
