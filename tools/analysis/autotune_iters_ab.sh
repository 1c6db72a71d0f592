# bench.py at the driver's protocol (20 steps, 5 warm-up), five runs each with the launch plans timed over 24 and over 64 iterations
set -e
cd $GRAFT_REPO_ROOT
for it in 24 64 24 64; do
  for i in 1 2 3; do
    ESPM_AUTOTUNE_ITERS=$it python bench.py --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('iters $it:', round(d['value']), d['config']['launch_plan'][:12])"
  done
done
