set -e
timeout -k 10 200 python tools/stereo_bench.py 10000 20
timeout -k 10 200 python tools/stereo_bench.py 100000 20
CCALIB_NO_FUSED=1 CCALIB_NO_AHEAD=1 timeout -k 10 200 python tools/stereo_bench.py 100000 20
