#!/bin/bash
# round 3, run ah: upload in chunks with the scans of X riding behind every chunk - estimator tests, whole fits
set -e
O=gpurun_out/r03ah; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_estimator.py tests/test_hyperspy_adapter.py tests/test_gpu_fullsize_parity.py tests/test_gpu_sharded_estimator.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing.log 2>&1 || { tail -30 $O/fit_timing.log; exit 1; }
grep "rep " $O/fit_timing.log; grep -A1 "rep 5" $O/fit_timing.log | tail -1 | cut -c1-600
