#!/bin/bash
# round 3, run v: a sampling thread inside the fits - where the main thread sits during the ~80 ms holes
set -e
O=gpurun_out/r03v; mkdir -p $O
SAMPLE=1 timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_sampled.log 2>&1 || { tail -30 $O/fit_timing_sampled.log; exit 1; }
grep -v amdgpu $O/fit_timing_sampled.log | cut -c1-330 | tail -120
