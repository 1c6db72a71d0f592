set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
timeout -k 10 300 python tools/analysis/w_finish_clock_c5.py 2>&1 | grep -v amdgpu
ROWS=128 python tools/analysis/c5_iter.py 2>&1 | tail -1
python tools/analysis/c5_iter.py 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -60 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
