echo "== pool off"; CCALIB_POOL_MB=0 timeout -k 10 300 python -u -m pytest tests/test_golden.py -m gpu -q --timeout 200 --timeout-method thread 2>&1 | tail -4
echo "== resume 4"; CCALIB_RESUME_LANES=4 timeout -k 10 300 python -u -m pytest tests/test_golden.py -m gpu -q --timeout 200 --timeout-method thread 2>&1 | tail -4
echo "== default"; timeout -k 10 300 python -u -m pytest tests/test_golden.py -m gpu -q --timeout 200 --timeout-method thread 2>&1 | tail -4
true
