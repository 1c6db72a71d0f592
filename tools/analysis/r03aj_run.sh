#!/bin/bash
# round 3, run aj: the whole GPU suite, the smoke test and the bench (driver's protocol) on the current tree
set -e
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1 || { tail -30 $O/bench_20_5.log; exit 1; }
tail -1 $O/bench_20_5.log | cut -c1-400
