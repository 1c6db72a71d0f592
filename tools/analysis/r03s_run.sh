#!/bin/bash
# round 3, run s: the whole-fit stall against the interpreter's garbage collector
set -e
O=gpurun_out/r03s; mkdir -p $O
L=tools/analysis
GC=log timeout -k 10 200 python $L/fit_timing.py > $O/fit_timing_gclog.log 2>&1 || { tail -30 $O/fit_timing_gclog.log; exit 1; }
GC=off timeout -k 10 200 python $L/fit_timing.py > $O/fit_timing_gcoff.log 2>&1 || { tail -30 $O/fit_timing_gcoff.log; exit 1; }
for f in gclog gcoff; do echo "== $f"; grep "fit_transform\|\[gc\]" $O/fit_timing_$f.log | cut -c1-300; done
