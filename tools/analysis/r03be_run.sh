#!/bin/bash
# round 3, run be: the packed wave stage in the two-launch H-step kernels as well - the whole GPU suite, BASELINE config 2's iteration
set -e
O=gpurun_out/r03be; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 200 python tools/analysis/c2_iter.py > $O/c2_iter.log 2>&1 || { tail -20 $O/c2_iter.log; exit 1; }
grep -v amdgpu $O/c2_iter.log | tail -6
