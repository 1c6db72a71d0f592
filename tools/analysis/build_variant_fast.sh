#!/bin/bash
# A variant of the narrow library that differs from the product's in a few translation units only: those are compiled with
# extra flags, the rest are the product's own objects (espm_amd/lib/*.o, built by __graft_entry__.build()).
#   bash tools/analysis/build_variant_fast.sh klprod "-DESPM_ELL_KLPROD=1" mu_fused mu_ell
# -> tools/analysis/libespm_mu_<name>.so, selected with ESPM_MU_LIB=<path> or tools/analysis/variant_ab.py.  Not the product.
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/analysis/variant_build_$NAME; mkdir -p $O
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/$f.o &
done
wait
OBJS=""
for o in $R/espm_amd/lib/mu_*.o; do
  b=$(basename $o .o)
  if [ -f $O/$b.o ]; then OBJS="$OBJS $O/$b.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/analysis/libespm_mu_$NAME.so $OBJS
rm -rf $O
ls -la $R/tools/analysis/libespm_mu_$NAME.so
