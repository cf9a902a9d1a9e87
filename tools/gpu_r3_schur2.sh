# Both dampings' Schur sums of a monocular pass in one launch (this build) against two launches (build/prev), same box:
# whole calls at several sizes, the LM loop at 1 000 and 100 000 views.
set -e
for v in 20 300 1000 10000 100000; do
  for rep in 1 2; do
    timeout -k 10 200 python3 tools/wholecall_bench.py $v | cut -c1-120
    CCALIB_OLD_LIB_OK=1 CCALIB_LIB=build/prev/libccalib.so timeout -k 10 200 python3 tools/wholecall_bench.py $v | cut -c1-120
  done
done
for v in 1000 100000; do
  timeout -k 10 300 python3 bench.py --no-cpu --views $v | python3 tools/bench_line.py
  CCALIB_OLD_LIB_OK=1 CCALIB_LIB=build/prev/libccalib.so timeout -k 10 300 python3 bench.py --no-cpu --views $v | python3 tools/bench_line.py
done
