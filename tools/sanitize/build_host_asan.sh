#!/bin/bash
# Host-instrumented build of the narrow library for the CPU-side sanitizer run:
# AddressSanitizer + UndefinedBehaviorSanitizer on the HOST half of every translation unit (argument validation, the exchange's
# context lifetime, the launchers' LDS / grid arithmetic); -fno-gpu-sanitize keeps the device code as the product builds it - GPU
# sanitizers are not available on this pool and never run on the GPU box.  Not the product: lives next to this script, git-ignored.
#   bash tools/sanitize/build_host_asan.sh && bash tools/sanitize/run_host_asan.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/sanitize/asan_build; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
J=${ESPM_BUILD_JOBS:-7}
pids=()
for f in mu_api mu_w_step mu_aux mu_ell mu_ell_build mu_l2 mu_fused mu_fused_plain mu_fused_stream mu_xchg mu_init; do
  /opt/rocm/bin/hipcc $FLAGS -c -I $R/include $R/espm_amd/csrc/$f.hip -o $O/$f.o &
  pids+=($!)
  while [ $(jobs -r | wc -l) -ge $J ]; do sleep 1; done
done
for i in 0 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DESPM_H_PARTS=4 -DESPM_H_PART=$i -c -I $R/include $R/espm_amd/csrc/mu_h_step.hip -o $O/mu_h_step_part$i.o &
  pids+=($!)
  while [ $(jobs -r | wc -l) -ge $J ]; do sleep 1; done
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o $R/tools/sanitize/libespm_mu_asan.so $O/*.o
rm -rf $O
ls -la $R/tools/sanitize/libespm_mu_asan.so
