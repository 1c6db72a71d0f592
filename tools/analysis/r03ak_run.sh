#!/bin/bash
# round 3, run ak: rehearsal of bench.py's N > 1 path on the one-GPU box (two and four ranks sharing the card, records through gloo)
set -e
O=gpurun_out/r03ak; mkdir -p $O
for n in 2 4; do
ESPM_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus $n --steps 20 --warmup 5 > $O/bench_gloo_$n.log 2>&1 || { tail -40 $O/bench_gloo_$n.log; exit 1; }
tail -1 $O/bench_gloo_$n.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config'].get('parallelism'), d['config'].get('record_exchange'))
for r in d['config']['per_rank']: print('   ', {k: r[k] for k in ('rank','rows','half_steps_us','w_step_with_exchange_us','lost_peers')})
"
done
