set -x
O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 600 python tools/analysis/fit_timing.py > $O/fit_timing.log 2>&1; grep "fit timing" $O/fit_timing.log | tail -3 | cut -c1-900; grep "^rank" $O/fit_timing.log | tail -4
ESPM_ENGINE_OVERLAP=0 timeout -k 10 600 python tools/analysis/fit_timing.py > $O/fit_timing_serial.log 2>&1; grep "^rank" $O/fit_timing_serial.log | tail -4
timeout -k 10 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -8 $O/t_all.log
