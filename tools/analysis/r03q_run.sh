#!/bin/bash
# round 3, run q: does the whole-fit stall follow the way the host waits for the device?  (interrupt-driven signal waits
# against polling, and the runtime's blocking synchronisation against spinning)
set -e
O=gpurun_out/r03q; mkdir -p $O
L=tools/analysis
timeout -k 10 200 python $L/fit_timing.py > $O/fit_timing_base.log 2>&1 || { tail -30 $O/fit_timing_base.log; exit 1; }
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python $L/fit_timing.py > $O/fit_timing_nointerrupt.log 2>&1 || { tail -30 $O/fit_timing_nointerrupt.log; exit 1; }
HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=2 timeout -k 10 200 python $L/fit_timing.py > $O/fit_timing_2queues.log 2>&1 || { tail -30 $O/fit_timing_2queues.log; exit 1; }
for f in base nointerrupt 2queues; do echo "== $f"; grep "fit_transform" $O/fit_timing_$f.log; done
