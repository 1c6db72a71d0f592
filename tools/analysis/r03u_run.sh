#!/bin/bash
# round 3, run u: the device's time line inside a whole fit (kernel + copy trace only), next to the host's section stamps
set -e
R=$(pwd); O=gpurun_out/r03u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace -- python3 $R/tools/analysis/fit_timing.py > $R/$O/fit_timing_traced.log 2>&1 || { tail -30 $R/$O/fit_timing_traced.log; exit 1; }
cd $R
python tools/analysis/fit_timeline.py $O/trace 0.5 > $O/fit_timeline.log 2>&1 || true
grep -A1 "rep 5" $O/fit_timing_traced.log | cut -c1-700
cat $O/fit_timeline.log
rm -rf $O/trace
