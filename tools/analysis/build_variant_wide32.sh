#!/bin/bash
# A variant of the 17..32-component library (libespm_mu_wide32.so) that differs from the product's in a few translation units: those are
# compiled with extra flags (mu_h_step in its four parts), the rest are the product's own objects (espm_amd/lib/wide32_*.o).
#   bash tools/analysis/build_variant_wide32.sh ct8 "-DESPM_MF_CT32=8" mu_w_step
# -> tools/analysis/libespm_mu_wide32_<name>.so, selected with ESPM_MU_WIDEST_LIB=<path>.  Not the product.
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/analysis/variant_build_w32_$NAME; mkdir -p $O
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DESPM_KP=32 -DESPM_MIN_K=17 -DESPM_MAX_K=32"
for f in "$@"; do
  if [ "$f" = "mu_h_step" ]; then
    for i in 0 1 2 3; do
      /opt/rocm/bin/hipcc $BASE $FLAGS -DESPM_H_PARTS=4 -DESPM_H_PART=$i -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/wide32_${f}_part$i.o &
    done
  else
    /opt/rocm/bin/hipcc $BASE $FLAGS -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/wide32_$f.o &
  fi
done
wait
OBJS=""
for o in $R/espm_amd/lib/wide32_mu_*.o; do
  b=$(basename $o)
  if [ -f $O/$b ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/analysis/libespm_mu_wide32_$NAME.so $OBJS
rm -rf $O
ls -la $R/tools/analysis/libespm_mu_wide32_$NAME.so
