set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -30 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
K="12 16" timeout -k 10 600 python tools/analysis/wide_iter.py 2>&1 | grep "u8"
