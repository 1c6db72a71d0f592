set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_estimator.py tests/test_hyperspy_adapter.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_est.log 2>&1 || (tail -40 $O/pytest_est.log | cut -c1-300; exit 1)
tail -1 $O/pytest_est.log
timeout -k 10 600 python tools/analysis/fit_profile.py > $O/fit_profile.log 2>&1
grep -v amdgpu.ids $O/fit_profile.log | cut -c1-160 | head -24
