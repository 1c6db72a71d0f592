#!/bin/bash
# A variant of the WIDE library (9..16 components) whose matrix-core translation units (mu_w_step, mu_h_step in its four parts)
# are compiled with extra flags; the rest are the product's objects (espm_amd/lib/wide_*.o).
#   bash tools/analysis/build_variant_wide.sh k16 "-DESPM_MFMA_K32_MASK=0"   -> tools/analysis/libespm_mu_wide_<name>.so (ESPM_MU_WIDE_LIB=<path>)
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/analysis/variant_build_wide_$NAME; mkdir -p $O
W="-DESPM_KP=16 -DESPM_MIN_K=9 -DESPM_MAX_K=16"
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $W $FLAGS -c -I $R/include"
$C $R/espm_amd/csrc/mu_w_step.hip -o $O/wide_mu_w_step.o &
for i in 0 1 2 3; do $C -DESPM_H_PARTS=4 -DESPM_H_PART=$i $R/espm_amd/csrc/mu_h_step.hip -o $O/wide_mu_h_step_part$i.o & done
wait
OBJS=""
for o in $R/espm_amd/lib/wide_mu_*.o; do b=$(basename $o); if [ -f $O/$b ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/analysis/libespm_mu_wide_$NAME.so $OBJS
rm -rf $O
ls -la $R/tools/analysis/libespm_mu_wide_$NAME.so
