# A/B of two library builds on the dominant kernel, same box: the in-tree library against $OLD (default build/old16)
set -e
OLD=${OLD:-$PWD/build/old16/libccalib.so}
for b in ${BOARDS:-9x6 8x6 7x6 6x5}; do
  echo "board $b new:"; BOARD=$b timeout -k 10 100 python tools/kernel_bench.py 100000 200
  echo "board $b old:"; BOARD=$b CCALIB_LIB=$OLD timeout -k 10 100 python tools/kernel_bench.py 100000 200
done
