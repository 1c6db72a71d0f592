set -e
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
tail -12 $O/fused_check.log
timeout -k 10 200 python tools/analysis/steps_sweep.py > $O/steps_sweep.log 2>&1; tail -14 $O/steps_sweep.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --no-cpu --steps 20 --warmup 5 > $R/$O/bench_20_5.log 2>&1
cd $R
python tools/analysis/trace_gaps.py "$O/trace/*/*_kernel_trace.csv" --iters 20 > $O/trace_gaps.log 2>&1; cat $O/trace_gaps.log
tail -1 $O/bench_20_5.log | cut -c1-300
rm -rf $O/trace
