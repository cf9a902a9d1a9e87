set -e
mkdir -p gpurun_out
rm -f gpurun_out/stereo_lm_trace.jsonl
CCALIB_TRACE=gpurun_out/stereo_lm_trace.jsonl CCALIB_TIMING=1 timeout -k 10 200 python3 tools/stereo_bench.py 100000 20 > gpurun_out/stereo_plain.log 2>&1
tail -2 gpurun_out/stereo_plain.log
