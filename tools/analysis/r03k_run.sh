set -e
O=gpurun_out/r03k; mkdir -p $O
L=tools/analysis
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for R in 1024 128; do
  for S in 1 0; do
    ROWS=$R ESPM_W_GSPLIT=$S timeout -k 10 200 python $L/c5_iter.py 2>&1 | grep "C5" | sed "s/^/gsplit=$S /" | tee -a $O/c5_iter.log
  done
done
timeout -k 10 300 python $L/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
grep "fused=\|FUSED_OK" $O/fused_check.log | tail -7
ROWS=64 timeout -k 10 200 python $L/shard_iter.py > $O/shard_iter_64.log 2>&1 || { tail -30 $O/shard_iter_64.log; exit 1; }
grep "C loop\|p2p        batch" $O/shard_iter_64.log
