set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
export CCALIB_NO_CHAIN=1 CCALIB_SPLIT_PUBLISH=1
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o st -- python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_prof.log 2>&1
python3 tools/rocpd_bygrid.py gpurun_out/prof/st_results.db colsum
python3 tools/rocpd_bygrid.py gpurun_out/prof/st_results.db publish
python3 tools/rocpd_gaps.py gpurun_out/prof/st_results.db 2>/dev/null | tail -25 || true
rm -rf gpurun_out/prof
