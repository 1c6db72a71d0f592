set -x
O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 120 ./tools/ubench/mfma_coissue > $O/mfma_coissue.log 2>&1
ESPM_BENCH_LOG=$PWD/$O/bench_4ranks_selflaunch.log timeout -k 10 1500 python -m pytest tests/test_gpu_bench_launch.py -x -q -m gpu > $O/t_launch.log 2>&1; tail -5 $O/t_launch.log
timeout -k 10 600 python -m pytest tests/test_gpu_estimator.py -x -q -m gpu -k "does_not_fit or physics" > $O/t_est.log 2>&1; tail -5 $O/t_est.log
timeout -k 10 600 python tools/analysis/fit_timing.py > $O/fit_timing.log 2>&1; grep -A1 "fit [2-4]:" $O/fit_timing.log | cut -c1-700
timeout -k 10 600 python tools/analysis/init_profile.py > $O/init_profile.log 2>&1; grep "^rep" $O/init_profile.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_20_5.log 2> $O/bench_20_5.err; tail -c 300 $O/bench_20_5.err
