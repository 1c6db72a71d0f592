set -e
O=gpurun_out/r02h; mkdir -p $O
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_20_5_$i.log 2>&1; tail -1 $O/bench_20_5_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20/5 autotuned:', round(d['value']), d['config']['launch_plan'], d['config']['launch_plan_timings_us'])"; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-autotune > $O/bench_20_5_na.log 2>&1; tail -1 $O/bench_20_5_na.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20/5 no autotune:', round(d['value']))"
timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default:', round(d['value']), 'steady', round(d['steady_state']['value']), d['loss_parity_rel'])"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
