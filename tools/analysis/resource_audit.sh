#!/bin/bash
# Registers, scratch and spills of every kernel of the narrow build, as the compiler reports them (-Rpass-analysis=kernel-resource-usage):
#   bash tools/analysis/resource_audit.sh [out.txt]
# Lists the kernels that use scratch memory or spill vector registers - a spilled register in a phase that is a chain of dependent
# round trips costs a round trip per reload (round 4: 92 spills at k = 8 were 22 us of a 111 us workgroup, NOTEBOOK.md section 9).
# Compiles every translation unit once more (objects under /tmp): ~3 minutes on 8 cores.  No GPU needed.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=${TMPDIR:-/tmp}/espm_resource_audit; mkdir -p $O
OUT=${1:-$R/profiles/resource_audit.txt}
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -I $R/include $2 $R/espm_amd/csrc/$1.hip -o $O/$3.o -Rpass-analysis=kernel-resource-usage 2> $O/$3.txt; }
run mu_w_step "" w & run mu_aux "" aux & run mu_ell "" ell & run mu_fused "" fused & run mu_fused_plain "" plain & run mu_fused_stream "" stream & run mu_l2 "" l2 &
wait
for i in 0 1 2 3; do run mu_h_step "-DESPM_H_PARTS=4 -DESPM_H_PART=$i" h$i & done
run mu_ell_build "" build & run mu_init "" init & run mu_xchg "" xchg &
wait
python3 - "$O" > "$OUT" <<'PY'
import re, glob, subprocess, shutil, sys
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
print("unit kernel | VGPRs scratch[B/lane] waves/SIMD SGPR-spills VGPR-spills   (only kernels with scratch or vector spills)")
n_all = 0
for f in sorted(glob.glob(sys.argv[1] + '/*.txt')):
    t = open(f).read()
    for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", t, flags=re.S):
        n_all += 1
        if int(m.group(4)) > 0 or int(m.group(7)) > 0:
            name = subprocess.run([filt, m.group(1)], capture_output=True, text=True).stdout.strip() if filt else m.group(1)
            name = re.sub(r"\(.*", "", name).replace("void espm::", "")
            print(f.split('/')[-1][:-4], name[:100], '|', m.group(3), m.group(4), m.group(5), m.group(6), m.group(7))
print(f"{n_all} kernels in all")
PY
echo "wrote $OUT"
