set -e
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_estimator.py -m gpu -x -q  > $O/pytest_wide.log 2>&1 || { tail -50 $O/pytest_wide.log; exit 1; }
tail -3 $O/pytest_wide.log
K="5 7 8 12 16" timeout -k 10 400 python tools/analysis/wide_iter.py > $O/wide_iter.log 2>&1; grep "k=" $O/wide_iter.log
