set -e
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hyperspy_adapter.py tests/test_gpu_fullsize_parity.py -m gpu -x -q -s > $O/pytest2.log 2>&1 || { tail -60 $O/pytest2.log; exit 1; }
grep "C3 five\|passed\|failed" $O/pytest2.log
timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1 || { tail -30 $O/bench_default.log; exit 1; }
tail -1 $O/bench_default.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_20_5.log 2>&1; tail -1 $O/bench_20_5.log | cut -c1-400
