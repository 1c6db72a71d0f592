#!/bin/bash
# round 3, run x: the whole fit against the container's CPU quota (cpu.max = 16 CPUs, 256 visible): thread pools sized to the
# visible cores burn the quota, the cgroup is throttled for the rest of the 100 ms period - the ~80 ms holes
set -e
O=gpurun_out/r03x; mkdir -p $O
for n in 16 8 4; do
  OMP_NUM_THREADS=$n OPENBLAS_NUM_THREADS=$n MKL_NUM_THREADS=$n timeout -k 10 200 python tools/analysis/fit_timing.py > $O/fit_timing_omp$n.log 2>&1 || { tail -30 $O/fit_timing_omp$n.log; exit 1; }
  echo "== OMP_NUM_THREADS=$n"; grep "rep \|cpu_count" $O/fit_timing_omp$n.log
done
grep -A1 "rep 5" $O/fit_timing_omp8.log | cut -c1-700
