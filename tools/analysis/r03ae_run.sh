#!/bin/bash
# round 3, run ae: full GPU suite on the current tree, then whole fits with the defaults
set -e
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing.log 2>&1 || { tail -30 $O/fit_timing.log; exit 1; }
grep "rep " $O/fit_timing.log
