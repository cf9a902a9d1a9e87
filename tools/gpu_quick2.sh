set -e
export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 60 rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1 || true
