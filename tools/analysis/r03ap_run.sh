#!/bin/bash
# round 3, run ap: the W accumulation on the 32-slot form with and without the one-wave-per-SIMD attribute (same box, alternating)
set -e
O=gpurun_out/r03ap; mkdir -p $O
for rep in 1 2; do for v in product noattr; do
  if [ $v = product ]; then unset ESPM_MU_WIDE_LIB; else export ESPM_MU_WIDE_LIB=$(pwd)/tools/analysis/libespm_mu_wide_$v.so; fi
  timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_${v}_$rep.log 2>&1 || { tail -20 $O/wide_repro_${v}_$rep.log; exit 1; }
  echo "== $v"; grep "run 2\|us / iteration" $O/wide_repro_${v}_$rep.log
done; done
