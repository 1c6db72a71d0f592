#!/bin/bash
# round 3, run at: kernel statistics of a 64-row and a 128-row shard's iteration (rocprofv3 --kernel-trace --stats on shard_iter.py)
set -e
R=$(pwd); O=gpurun_out/r03at; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rows in 64 128; do
  ROWS=$rows timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks$rows -- python3 $R/tools/analysis/shard_iter.py > $R/$O/shard_iter_$rows.log 2>&1 || { tail -20 $R/$O/shard_iter_$rows.log; exit 1; }
  for f in $R/$O/ks$rows/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $R/$O/ks${rows}_kernel_stats.csv; done
  rm -rf $R/$O/ks$rows
  echo "rows $rows"; cut -d, -f1-4 $R/$O/ks${rows}_kernel_stats.csv | cut -c1-150 | head -8
done
