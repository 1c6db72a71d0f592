set -e
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python tools/analysis/default_args_iter.py > $O/default_args_iter.log 2>&1; cat $O/default_args_iter.log | grep -v amdgpu.ids
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1; grep "fused=\|max |dW\|FUSED_OK\|Error\|error" $O/fused_check.log | tail -8
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --no-cpu --steps 300 --warmup 30 > $R/$O/bench_300.log 2>&1
cd $R
python tools/analysis/trace_gaps.py "$O/trace/*/*_kernel_trace.csv" --iters 200 > $O/trace_gaps.log 2>&1; cat $O/trace_gaps.log
tail -1 $O/bench_300.log | cut -c1-200
rm -rf $O/trace
