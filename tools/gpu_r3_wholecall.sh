# Whole calibrateCamera / stereoCalibrate calls from host buffers: phase times, the three call shapes, the new entry
# points' tests, then the complete GPU suite.
set -e
CCALIB_TIMING=1 timeout -k 10 200 python tools/wholecall_bench.py 100000 2> gpurun_out/wc_phases.txt
tail -40 gpurun_out/wc_phases.txt
timeout -k 10 200 python tools/wholecall_bench.py stereo 10000
timeout -k 10 300 python -u -m pytest tests/test_gpu_views_entry.py -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/t8.log 2>&1 || { tail -60 gpurun_out/t8.log; exit 1; }
tail -3 gpurun_out/t8.log
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t9.log 2>&1 || { tail -60 gpurun_out/t9.log; exit 1; }
tail -3 gpurun_out/t9.log
