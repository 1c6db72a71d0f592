set -e
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python tools/analysis/default_args_iter.py > $O/default_args_iter.log 2>&1; cat $O/default_args_iter.log | grep -v amdgpu.ids
