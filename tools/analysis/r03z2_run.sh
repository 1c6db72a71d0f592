#!/bin/bash
# round 3, run z2: the 32-slot matrix instruction in the W accumulation only (both of its sites) against the 16-slot build: time and reproducibility
set -e
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2; do for m in 0 12; do
  ESPM_MU_WIDE_LIB=$(pwd)/tools/analysis/libespm_mu_wide_m$m.so timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_m${m}_$rep.log 2>&1 || { tail -20 $O/wide_repro_m${m}_$rep.log; exit 1; }
  echo "== mask $m"; grep "run \|us / iteration" $O/wide_repro_m${m}_$rep.log
done; done
for k in 9 12; do for m in 0 12; do
  K=$k ESPM_MU_WIDE_LIB=$(pwd)/tools/analysis/libespm_mu_wide_m$m.so timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_k${k}_m${m}.log 2>&1 || { tail -20 $O/wide_repro_k${k}_m${m}.log; exit 1; }
  echo "== k $k mask $m"; grep "run \|us / iteration" $O/wide_repro_k${k}_m${m}.log
done; done
