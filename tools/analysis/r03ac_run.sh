#!/bin/bash
# round 3, run ac: with the thread pools inside the CPU budget - the host copy of X_ started right after the upload against at the loop
set -e
O=gpurun_out/r03ac; mkdir -p $O
for rep in 1 2; do for m in loop early; do
  ESPM_HOSTCOPY_START=$m timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_${m}_$rep.log 2>&1 || { tail -30 $O/fit_timing_${m}_$rep.log; exit 1; }
  echo "== $m"; grep "rep " $O/fit_timing_${m}_$rep.log
done; done
grep -A1 "rep 5" $O/fit_timing_early_2.log | cut -c1-700
