set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -40 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
python tools/analysis/default_args_iter.py 2>&1 | grep "us/iteration" > $O/default_args_iter.log; cat $O/default_args_iter.log
ROWS=128 python tools/analysis/c5_iter.py 2>&1 | tail -1
python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | cut -c1-200
