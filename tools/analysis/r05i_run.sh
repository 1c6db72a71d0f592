set -x
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -4 $O/t_all.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_20_5.log 2> $O/bench_20_5.err; cut -c1-200 $O/bench_20_5.log
timeout -k 10 600 python bench.py --no-cpu --no-extras > $O/bench_300.log 2> $O/bench_300.err; cut -c1-200 $O/bench_300.log
timeout -k 10 600 python tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1; tail -1 $O/c5_iter.log
CONFIG=c5 ROWS=128 timeout -k 10 600 python tools/analysis/shard_iter.py > $O/shard_iter_c5_128.log 2>&1; grep "us/it" $O/shard_iter_c5_128.log
for RW in 64 128; do ROWS=$RW timeout -k 10 600 python tools/analysis/shard_iter.py > $O/shard_iter_$RW.log 2>&1; grep "us/it" $O/shard_iter_$RW.log | head -3; done
for KK in 3 6 7; do K=$KK ROWS=512 timeout -k 10 300 python tools/analysis/small_iter.py > $O/k${KK}_iter.log 2>&1; tail -1 $O/k${KK}_iter.log; done
