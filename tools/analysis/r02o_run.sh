# scratch script of the last verification run of round 2 (GPU box): full GPU suite, smoke, bench at the driver's protocol and by default
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -40 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1
tail -1 $O/bench_20_5.log | cut -c1-260
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-260
