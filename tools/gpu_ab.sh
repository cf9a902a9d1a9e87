# A/B of two library builds on the dominant kernel: the in-tree library against build/old16/libccalib.so
set -e
for b in 9x6 8x6 7x6 6x5; do
  echo "board $b new:"; BOARD=$b timeout -k 10 100 python tools/kernel_bench.py 100000 200
  echo "board $b old:"; BOARD=$b CCALIB_LIB=$PWD/build/old16/libccalib.so timeout -k 10 100 python tools/kernel_bench.py 100000 200
done
