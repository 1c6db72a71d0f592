set -e
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
grep "fused=\|max |dW\|FUSED_OK" $O/fused_check.log
timeout -k 10 300 python tools/analysis/phase_clock.py > $O/phase_clock.log 2>&1 || { tail -30 $O/phase_clock.log; exit 1; }
tail -12 $O/phase_clock.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
