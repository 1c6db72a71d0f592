#!/bin/bash
# A build of the narrow library with extra compiler flags for A/B runs: tools/analysis/libespm_mu_<name>.so, selected with
# ESPM_MU_LIB=<path>.  Not the product: git-ignored, travels to the GPU box with gpurun.
#   bash tools/analysis/build_variant_lib.sh t512pf2 "-DESPM_FUSED_SMALL_THREADS=512 -DESPM_FUSED_SMALL_PREFETCH=2"
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/analysis/variant_build_$NAME; mkdir -p $O
for f in mu_api mu_h_step mu_w_step mu_aux mu_ell mu_ell_build mu_l2 mu_fused mu_fused_plain mu_fused_stream mu_xchg; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/analysis/libespm_mu_$NAME.so $O/*.o
rm -rf $O
ls -la $R/tools/analysis/libespm_mu_$NAME.so
