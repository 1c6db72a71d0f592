#!/bin/bash
# round 3, run aa: the wide build with the 32-slot matrix instruction in the W accumulation - its tests, reproducibility, time
set -e
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_estimator.py tests/test_gpu_updates.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for k in 16 9; do
  K=$k timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_k$k.log 2>&1 || { tail -20 $O/wide_repro_k$k.log; exit 1; }
  grep "run \|us / iteration" $O/wide_repro_k$k.log
done
