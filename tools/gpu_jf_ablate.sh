# Timing-only builds of the fused Jacobian kernel with one part removed each (build/jfx1..4, see CCALIB_JF_ABLATE
# in csrc/kernels.hip): what the parts cost when the others are still there.
set -e
echo "full:"; timeout -k 10 100 python tools/kernel_bench.py 100000 100
for x in 1 2 3 4; do echo "ablate $x:"; CCALIB_LIB=$PWD/build/jfx$x/libccalib.so timeout -k 10 100 python tools/kernel_bench.py 100000 100; done
