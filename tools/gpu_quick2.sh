set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 --timeout-method thread -k "extreme" 2>&1 | tail -40
