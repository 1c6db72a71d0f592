#!/bin/bash
# round 3, run bd: the wave stage of the fused kernel's record reduction through half / row exchanges (wave_reduce_packed) - tests, A/B
set -e
O=gpurun_out/r03bd; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_updates.py tests/test_gpu_estimator.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rows in 512 64; do
  ROWS=$rows timeout -k 10 300 python tools/analysis/variant_ab.py before=tools/analysis/libespm_mu_before.so now=espm_amd/lib/libespm_mu.so > $O/ab_$rows.log 2>&1 || { tail -20 $O/ab_$rows.log; }
  echo "rows $rows: $(grep 'best\|rel dloss' $O/ab_$rows.log | tr '\n' ' ')"
done
