set -e
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -25 $O/pytest.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1 || { tail -30 $O/bench_20_5.log; exit 1; }
tail -1 $O/bench_20_5.log | cut -c1-1500
ESPM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/bench_2ranks_gloo.log 2>&1 || { tail -30 $O/bench_2ranks_gloo.log; exit 1; }
tail -1 $O/bench_2ranks_gloo.log | cut -c1-2500
timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1 || { tail -30 $O/shard_iter.log; exit 1; }
grep -v amdgpu $O/shard_iter.log
