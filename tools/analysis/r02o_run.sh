# scratch script of the last verification run of round 2 (GPU box)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -60 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
python tools/analysis/default_args_iter.py 2>&1 | grep "us/iteration" > $O/default_args_iter.log; cat $O/default_args_iter.log
K=8 python tools/analysis/default_args_iter.py 2>&1 | grep "reference default"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1
tail -1 $O/bench_20_5.log | cut -c1-200
