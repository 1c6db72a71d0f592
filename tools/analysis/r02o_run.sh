# scratch script of the last verification run of round 2 (GPU box): bench logs with the final code
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O
cd $R
python bench.py > $O/bench_default.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1
tail -1 $O/bench_default.log | cut -c1-160; tail -1 $O/bench_20_5.log | cut -c1-160
