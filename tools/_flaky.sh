for i in 1 2 3 4 5 6 7 8 9 10; do
  python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | cut -c1-200
done
