set -e
O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
K="9 12" timeout -k 10 400 python tools/analysis/wide_iter.py > $O/wide_iter.log 2>&1; grep "k=" $O/wide_iter.log
