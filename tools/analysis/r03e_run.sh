set -e
O=gpurun_out/r03e; mkdir -p $O
L=tools/analysis
timeout -k 10 300 python -m pytest tests/test_gpu_estimator.py -m gpu -q -k "simplex_over_w" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for R in 64 128 256; do
  ROWS=$R timeout -k 10 200 python $L/shard_iter.py > $O/shard_iter_$R.log 2>&1 || { tail -30 $O/shard_iter_$R.log; exit 1; }
  echo "rows $R"; grep "us/it\|lost" $O/shard_iter_$R.log
done
ROWS=64 FUSED=1 timeout -k 10 200 python $L/phase_clock.py > $O/phase_clock_64rows.log 2>&1 || { tail -30 $O/phase_clock_64rows.log; exit 1; }
grep -v amdgpu $O/phase_clock_64rows.log
timeout -k 10 200 python $L/c2_iter.py > $O/c2_iter.log 2>&1 || { tail $O/c2_iter.log; exit 1; }
ESPM_FUSED=0 timeout -k 10 200 python $L/c2_iter.py >> $O/c2_iter.log 2>&1 || { tail $O/c2_iter.log; exit 1; }
grep "C2" $O/c2_iter.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1 || { tail -30 $O/bench_20_5.log; exit 1; }
tail -1 $O/bench_20_5.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'], d.get('steady_state'), d.get('product_default'), d.get('c5'), d['cpu_baseline']['value'], d['cpu_baseline'].get('full_size'), d['loss_parity_rel'])"

O=gpurun_out/r03f; mkdir -p $O
R=$PWD
timeout -k 10 300 python tools/analysis/fit_phases.py > $O/fit_phases.log 2>&1 || { tail -30 $O/fit_phases.log; exit 1; }
grep -v amdgpu $O/fit_phases.log | tail -40
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/fit_trace -- python3 $R/tools/analysis/fit_phases.py > $R/$O/fit_phases_traced.log 2>&1 || { tail -30 $R/$O/fit_phases_traced.log; exit 1; }
cd $R
python tools/analysis/fit_trace_summary.py $O/fit_trace > $O/fit_trace_summary.log 2>&1 || true
cat $O/fit_trace_summary.log
grep -v amdgpu $O/fit_phases_traced.log | tail -14
ls $O/fit_trace/*/ | head; du -sh $O/fit_trace
rm -rf $O/fit_trace
