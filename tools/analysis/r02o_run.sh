set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -60 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
ROWS=128 python tools/analysis/c5_iter.py 2>&1 | tail -1
python tools/analysis/c5_iter.py 2>&1 | tail -1
