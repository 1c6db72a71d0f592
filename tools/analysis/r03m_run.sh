set -e
O=gpurun_out/r03m; mkdir -p $O
L=tools/analysis
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
CONFIG=c5 ROWS=128 timeout -k 10 300 python $L/shard_iter.py > $O/shard_iter_c5_128.log 2>&1 || { tail -30 $O/shard_iter_c5_128.log; exit 1; }
grep "us/it\|lost" $O/shard_iter_c5_128.log
CONFIG=c5 ROWS=128 ESPM_W_GSPLIT=0 timeout -k 10 300 python $L/shard_iter.py > $O/shard_iter_c5_128_nosplit.log 2>&1 || { tail -30 $O/shard_iter_c5_128_nosplit.log; exit 1; }
grep "us/it" $O/shard_iter_c5_128_nosplit.log
