#!/bin/bash
# round 3, run ba: the record reduction as one workgroup per value in every launch that carries it - the whole GPU suite, shards, bench
set -e
O=gpurun_out/r03ba; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rows in 64 128 256; do
  ROWS=$rows timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_$rows.log 2>&1 || { tail -20 $O/shard_iter_$rows.log; exit 1; }
  echo "rows $rows: $(grep 'unsharded C loop' $O/shard_iter_$rows.log | cut -c32-) | $(grep 'p2p        batch' $O/shard_iter_$rows.log | cut -c32-)"
done
SIMPLEX_W=1 ROWS=64 timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_simplexw_64.log 2>&1; echo "simplex W, 64 rows: $(grep 'unsharded C loop' $O/shard_simplexw_64.log | cut -c32-) | $(grep 'p2p        batch' $O/shard_simplexw_64.log | cut -c32-)"
CONFIG=c5 ROWS=128 timeout -k 10 300 python tools/analysis/shard_iter.py > $O/shard_c5_128.log 2>&1; echo "C5 128 rows: $(grep 'unsharded C loop' $O/shard_c5_128.log | cut -c32-) | $(grep 'p2p        batch' $O/shard_c5_128.log | cut -c32-)"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1 || { tail -30 $O/bench_20_5.log; exit 1; }
tail -1 $O/bench_20_5.log | cut -c1-200
