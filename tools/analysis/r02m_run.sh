set -e
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
grep "fused=\|max |dW\|FUSED_OK" $O/fused_check.log | tail -8
COUNTS=18 timeout -k 10 300 python tools/analysis/phase_clock.py > $O/phase_clock_18.log 2>&1; grep "kernel span\|H walk (wave\|W walk (wave" $O/phase_clock_18.log
timeout -k 10 300 python tools/analysis/phase_clock.py > $O/phase_clock_500.log 2>&1; grep "kernel span\|H walk\|W walk\|table" $O/phase_clock_500.log
timeout -k 10 400 python tools/analysis/dose_iter.py > $O/dose_iter.log 2>&1; grep "N=" $O/dose_iter.log
