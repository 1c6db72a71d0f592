set -e
O=gpurun_out/r02k; mkdir -p $O
ESPM_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo.log 2>&1 || { tail -30 $O/bench_2ranks_gloo.log; exit 1; }
tail -1 $O/bench_2ranks_gloo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss_first'], d['loss_last'], d['config']['record_exchange'])"
ESPM_XCHG=collective ESPM_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_coll.log 2>&1 || { tail -30 $O/bench_2ranks_gloo_coll.log; exit 1; }
tail -1 $O/bench_2ranks_gloo_coll.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss_first'], d['loss_last'], d['config']['record_exchange'])"
