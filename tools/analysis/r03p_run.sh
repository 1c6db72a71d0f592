#!/bin/bash
# round 3, run p: the sharded simplex-over-W update through the granule exchange (two launches) - tests, then a rank's iteration
set -e
mkdir -p gpurun_out/r03p
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_estimator.py tests/test_gpu_sharded_fullsize.py -m gpu -x -q > gpurun_out/r03p/pytest.log 2>&1
for rows in 64 128; do
  SIMPLEX_W=1 ROWS=$rows timeout -k 10 300 python tools/analysis/shard_iter.py > gpurun_out/r03p/shard_simplexw_$rows.log 2>&1
done
tail -3 gpurun_out/r03p/pytest.log; cat gpurun_out/r03p/shard_simplexw_64.log gpurun_out/r03p/shard_simplexw_128.log
