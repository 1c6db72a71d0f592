#!/bin/bash
# round 3, run ag: the LU normaliser kernel and the Cholesky QR of the device initialisation - tests, sections, whole fits
set -e
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_updates.py tests/test_gpu_estimator.py -m gpu -x -q -k "lu_normaliser or nndsvd or golden or freed" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/analysis/init_gemm_time.py > $O/init_sections.log 2>&1 || { tail -30 $O/init_sections.log; exit 1; }
grep -v amdgpu $O/init_sections.log | tail -14
timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing.log 2>&1 || { tail -30 $O/fit_timing.log; exit 1; }
grep "rep " $O/fit_timing.log; grep -A1 "rep 5" $O/fit_timing.log | tail -1 | cut -c1-600
