#!/bin/bash
# round 3, run as: list dwords per batch (2 instead of 4: about half the walks' code) in the run-time-sized fused instance - shards of 64 and 128 rows
set -e
O=gpurun_out/r03as; mkdir -p $O
for rows in 64 128; do for rep in 1 2; do for v in product u22 u42 u24; do
  if [ $v = product ]; then unset ESPM_MU_LIB; else export ESPM_MU_LIB=$(pwd)/tools/analysis/libespm_mu_$v.so; fi
  ROWS=$rows timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_${rows}_${v}_$rep.log 2>&1 || { tail -20 $O/shard_${rows}_${v}_$rep.log; exit 1; }
  echo "rows $rows $v: $(grep 'unsharded C loop' $O/shard_${rows}_${v}_$rep.log | cut -c32-) | $(grep 'p2p        batch' $O/shard_${rows}_${v}_$rep.log | cut -c32-)"
done; done; done
