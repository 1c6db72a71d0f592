#!/bin/bash
# round 3, run z: which of the four matrix-instruction sites of the wide build loses run-to-run reproducibility with the 32-slot form
set -e
O=gpurun_out/r03z; mkdir -p $O
for m in 1 2 4 8; do
  ESPM_MU_WIDE_LIB=$(pwd)/tools/analysis/libespm_mu_wide_m$m.so timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_m$m.log 2>&1 || { tail -20 $O/wide_repro_m$m.log; exit 1; }
  echo "== mask $m"; grep "run \|us / iteration" $O/wide_repro_m$m.log
done
