set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in 8 16 64; do
  for cut in 4 5 6; do
    export CCALIB_INIT_CUT=$cut CCALIB_RESUME_LANES=$L
    D=$R/gpurun_out/rg_${L}_$cut
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o wc -- python3 $R/tools/wholecall_bench.py 100000 > $R/gpurun_out/rf.log 2>&1
    echo "L=$L cut=$cut"
    grep views $R/gpurun_out/rf.log | cut -c1-120
  done
done
