set -e
O=gpurun_out/r03e; mkdir -p $O
L=tools/analysis
for v in hold0 hold1 hold3 hold7; do
  ESPM_MU_WIDE_LIB=$PWD/$L/libespm_mu_wide_$v.so timeout -k 10 200 python $L/wide_repro.py 2>&1 | grep -v amdgpu | tee -a $O/wide_repro.log
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect "tests/test_gpu_fullsize.py::test_matrix_core_kernels_of_the_wide_build_at_full_size" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
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
