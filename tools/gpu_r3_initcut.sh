set -e
cd /tmp && export TMPDIR=/tmp
for cut in 1 3 6 20; do
  export CCALIB_INIT_CUT=$cut
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cut$cut -o wc -- python3 $GRAFT_REPO_ROOT/tools/wholecall_bench.py 100000 > $GRAFT_REPO_ROOT/gpurun_out/cut$cut.log 2>&1
  echo "cut $cut"; grep -h "k_init" $GRAFT_REPO_ROOT/gpurun_out/cut$cut/wc_kernel_stats.csv
done
