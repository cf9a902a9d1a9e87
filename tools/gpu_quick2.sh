set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_stereo
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stereo -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
tail -1 gpurun_out/stereo_prof.log
