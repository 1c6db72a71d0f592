set -e
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1 || { tail -30 $O/shard_iter.log; exit 1; }
grep "us/it\|lost" $O/shard_iter.log
ESPM_FUSED=always timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_fused.log 2>&1 || { tail -30 $O/shard_iter_fused.log; exit 1; }
grep "us/it" $O/shard_iter_fused.log
ROWS=128 ESPM_FUSED=always timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_fused_128.log 2>&1 || { tail -30 $O/shard_iter_fused_128.log; exit 1; }
grep "us/it" $O/shard_iter_fused_128.log
for R in 64; do
  ROWS=$R FUSED=always timeout -k 10 200 python tools/analysis/phase_clock.py > $O/phase_clock_${R}rows.log 2>&1 || { tail -30 $O/phase_clock_${R}rows.log; exit 1; }
  grep -v amdgpu $O/phase_clock_${R}rows.log
done
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
grep "fused=\|max |dW\|FUSED_OK" $O/fused_check.log | tail -8
K="12 16" timeout -k 10 300 python tools/analysis/wide_iter.py > $O/wide_iter.log 2>&1 || { tail -30 $O/wide_iter.log; exit 1; }
grep "k=" $O/wide_iter.log
