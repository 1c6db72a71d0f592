set -e
O=gpurun_out/r03g; mkdir -p $O
L=tools/analysis
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for R in 64 128 256; do
  ROWS=$R timeout -k 10 200 python $L/shard_iter.py > $O/shard_iter_$R.log 2>&1 || { tail -30 $O/shard_iter_$R.log; exit 1; }
  echo "rows $R"; grep "us/it\|lost" $O/shard_iter_$R.log
done
ROWS=64 FUSED=1 timeout -k 10 200 python $L/phase_clock.py > $O/phase_clock_64rows.log 2>&1 || { tail -30 $O/phase_clock_64rows.log; exit 1; }
grep -v amdgpu $O/phase_clock_64rows.log
timeout -k 10 200 python $L/c2_iter.py > $O/c2_iter.log 2>&1 || { tail $O/c2_iter.log; exit 1; }
grep "C2" $O/c2_iter.log
timeout -k 10 100 python $L/host_fault_probe.py > $O/host_fault_probe.log 2>&1 || true
cat $O/host_fault_probe.log
NOCOPY=1 timeout -k 10 300 python $L/fit_phases.py > $O/fit_phases_nocopy.log 2>&1 || { tail -30 $O/fit_phases_nocopy.log; exit 1; }
grep "rep \|unaccounted\|upload\|initialize\|engine set\|iterate" $O/fit_phases_nocopy.log
