#!/bin/bash
# round 3, run ab / an: the 32-slot form in the H-step kernel - the SAME code with two workgroups per CU (h3two) and with one
# (h3pad: 24 KB of LDS requested that nobody uses); earlier: without the scheduling fence, and built for one workgroup per CU
set -e
O=gpurun_out/r03an; mkdir -p $O
for v in h3two h3pad h3two h3pad; do
  ESPM_MU_WIDE_LIB=$(pwd)/tools/analysis/libespm_mu_wide_$v.so timeout -k 10 200 python tools/analysis/wide_repro.py > $O/wide_repro_$v.log 2>&1 || { tail -20 $O/wide_repro_$v.log; exit 1; }
  echo "== $v"; grep "run \|us / iteration" $O/wide_repro_$v.log
done
