#!/bin/bash
# Instrumented build of the narrow library (-DESPM_PHASE_CLOCK: phase stamps in the fused kernel) for phase_clock.py.
# Not the product: lives next to this script, is git-ignored, travels to the GPU box with gpurun.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/analysis/phase_build; mkdir -p $O
for f in mu_api mu_h_step mu_w_step mu_aux mu_ell mu_ell_build mu_l2 mu_fused mu_fused_plain mu_fused_stream mu_xchg mu_init; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DESPM_PHASE_CLOCK -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/analysis/libespm_mu_phase.so $O/*.o
rm -rf $O
ls -la $R/tools/analysis/libespm_mu_phase.so
