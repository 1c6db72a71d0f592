set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O
cd $R
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20/5:', round(d['value']), d['config']['launch_plan'][:14], d['config']['launch_plan_timings_us'])"; done
python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default:', round(d['value']), d.get('steady_state',{}).get('value'))"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -1
