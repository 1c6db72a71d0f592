#!/bin/bash
# round 3, run av: timing-only - the exchange launch without its wait for the statistics' granules and the halo flags
set -e
O=gpurun_out/r03av; mkdir -p $O
for rep in 1 2 3; do for v in product nostat; do
  if [ $v = product ]; then unset ESPM_MU_LIB; else export ESPM_MU_LIB=$(pwd)/tools/analysis/libespm_mu_$v.so; fi
  ROWS=64 timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_${v}_$rep.log 2>&1 || { tail -20 $O/shard_${v}_$rep.log; exit 1; }
  echo "== $v: $(grep 'p2p        batch' $O/shard_${v}_$rep.log)"
done; done
