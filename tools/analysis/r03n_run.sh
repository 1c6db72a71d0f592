O=gpurun_out/r03n; mkdir -p $O
L=tools/analysis
for i in 1 2 3; do
  timeout -k 10 300 python -m pytest tests/test_gpu_sharded_estimator.py -m gpu -q -k "pg" > $O/pytest_pg_$i.log 2>&1; echo "run $i rc=$?"; tail -1 $O/pytest_pg_$i.log
done
ESPM_W_GSPLIT=0 timeout -k 10 300 python -m pytest tests/test_gpu_sharded_estimator.py -m gpu -q -k "pg" > $O/pytest_pg_nosplit.log 2>&1; echo "nosplit rc=$?"; tail -1 $O/pytest_pg_nosplit.log
CONFIG=c5 ROWS=128 timeout -k 10 300 python $L/shard_iter.py > $O/shard_iter_c5_128.log 2>&1 || { tail -30 $O/shard_iter_c5_128.log; exit 1; }
grep "us/it\|lost" $O/shard_iter_c5_128.log
CONFIG=c5 ROWS=128 ESPM_W_GSPLIT=0 timeout -k 10 300 python $L/shard_iter.py > $O/shard_iter_c5_128_nosplit.log 2>&1 || { tail -30 $O/shard_iter_c5_128_nosplit.log; exit 1; }
grep "us/it" $O/shard_iter_c5_128_nosplit.log
