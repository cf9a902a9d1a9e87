set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_whole
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_whole -o whole -- python3 tools/wholecall_bench.py 100000 > gpurun_out/whole_prof.log 2>&1
