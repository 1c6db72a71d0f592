set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -40 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
ROWS=16,32,64,128,256 python tools/analysis/small_iter.py > $O/small_default.log 2>&1
python tools/analysis/c2_iter.py > $O/c2.log 2>&1
python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1
python bench.py > $O/bench_default.log 2>&1
cat $O/small_default.log $O/c2.log | grep -v amdgpu.ids
grep "us/it" $O/shard_iter.log
tail -1 $O/bench_default.log | cut -c1-400
