set -x
O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_fullsize.py tests/test_gpu_fullsize_parity.py -x -q -m gpu -k "config5 or c5" > $O/t_c5.log 2>&1; tail -3 $O/t_c5.log
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_sharded_estimator.py -x -q -m gpu > $O/t_fuzz.log 2>&1; tail -3 $O/t_fuzz.log
timeout -k 10 600 python tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1; tail -1 $O/c5_iter.log
ESPM_W_GCOL=0 timeout -k 10 600 python tools/analysis/c5_iter.py > $O/c5_iter_gcol0.log 2>&1; tail -1 $O/c5_iter_gcol0.log
CONFIG=c5 ROWS=128 timeout -k 10 600 python tools/analysis/shard_iter.py > $O/shard_iter_c5_128.log 2>&1; cat $O/shard_iter_c5_128.log
CONFIG=c5 ROWS=128 ESPM_W_GCOL=0 timeout -k 10 600 python tools/analysis/shard_iter.py > $O/shard_iter_c5_128_gcol0.log 2>&1; cat $O/shard_iter_c5_128_gcol0.log
