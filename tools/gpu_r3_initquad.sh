# Homography kernels with four lanes per view (in-tree) against the build before (build/prev): kernel times under the
# profiler, whole calls at 100 000 / 1 000 / 20 views, then the initialisation and parity tests.
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for tag in new old; do
  if [ $tag = old ]; then export CCALIB_LIB=$R/build/prev/libccalib.so CCALIB_OLD_LIB_OK=1; fi
  rm -rf gpurun_out/iq_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/iq_$tag -o wc -- python3 tools/wholecall_bench.py 100000 > gpurun_out/iq_$tag.log 2>&1
  echo "$tag:"; grep views gpurun_out/iq_$tag.log | cut -c1-300
  for n in 1000 20; do timeout -k 10 100 python3 tools/wholecall_bench.py $n | cut -c1-300; done
done
unset CCALIB_LIB CCALIB_OLD_LIB_OK
timeout -k 10 900 python -u -m pytest tests/test_gpu_init.py tests/test_golden.py tests/test_lm_trace.py tests/test_gpu_parity.py tests/test_gpu_views_entry.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/t14.log 2>&1 || { tail -60 gpurun_out/t14.log; exit 1; }
tail -3 gpurun_out/t14.log
