# Same-box A/B of the bench line (default 20 steps, stereo block included): the in-tree library against $OLD, alternating
set -e
OLD=${OLD:-$PWD/build/prev/libccalib.so}
for i in 1 2; do
  echo -n "new: "; timeout -k 10 300 python bench.py --cpu-views 200 2>/dev/null | python3 tools/bench_line.py
  echo -n "old: "; CCALIB_LIB=$OLD timeout -k 10 300 python bench.py --cpu-views 200 2>/dev/null | python3 tools/bench_line.py
done
