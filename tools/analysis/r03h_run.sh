set -e
O=gpurun_out/r03h; mkdir -p $O
L=tools/analysis
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_estimator.py tests/test_gpu_sharded_fullsize.py -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for R in 64 128 256; do
  ROWS=$R timeout -k 10 200 python $L/shard_iter.py > $O/shard_iter_$R.log 2>&1 || { tail -30 $O/shard_iter_$R.log; exit 1; }
  echo "rows $R"; grep "C loop\|p2p        batch" $O/shard_iter_$R.log
done
timeout -k 10 300 python $L/fit_timing.py > $O/fit_timing.log 2>&1 || { tail -30 $O/fit_timing.log; exit 1; }
grep -v amdgpu $O/fit_timing.log
