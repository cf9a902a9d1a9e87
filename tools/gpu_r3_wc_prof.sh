set -e
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/wcprof -o wc -- python3 $GRAFT_REPO_ROOT/tools/wholecall_bench.py 100000 > $GRAFT_REPO_ROOT/gpurun_out/wcprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/wcprof | head
