# runs the GPU tests whose names match $1 (pytest -k), showing failures in full
set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -k "$1" 2>&1 | tail -60
