#!/bin/bash
# round 3, run y: whole fits with the product's own thread-pool limit (no environment variables), both upload paths; GPU tests; bench
set -e
O=gpurun_out/r03y; mkdir -p $O
ESPM_UPLOAD=staged timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_staged.log 2>&1 || { tail -30 $O/fit_timing_staged.log; exit 1; }
ESPM_UPLOAD=plain timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_plain.log 2>&1 || { tail -30 $O/fit_timing_plain.log; exit 1; }
ESPM_CPU_THREADS=0 timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_unlimited.log 2>&1 || { tail -30 $O/fit_timing_unlimited.log; exit 1; }
for f in staged plain unlimited; do echo "== $f"; grep "rep " $O/fit_timing_$f.log; done
grep -A1 "rep 5" $O/fit_timing_staged.log | cut -c1-700
grep -A1 "rep 5" $O/fit_timing_plain.log | cut -c1-700
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1 || { tail -30 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-1500
