#!/bin/bash
# round 3, run au: the GPU suite three times in a row (looking for anything intermittent)
set -e
O=gpurun_out/r03au; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 380 python -m pytest tests -m gpu -q > $O/pytest_$i.log 2>&1 || { tail -60 $O/pytest_$i.log; exit 1; }
  tail -1 $O/pytest_$i.log
done
