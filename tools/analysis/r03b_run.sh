set -e
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_fullsize.py tests/test_hyperspy_adapter.py -m gpu -x -q --durations=5 > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -25 $O/pytest.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2>&1 || { tail -30 $O/bench_20_5.log; exit 1; }
tail -1 $O/bench_20_5.log | cut -c1-1500
ESPM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/bench_2ranks_gloo.log 2>&1 || { tail -30 $O/bench_2ranks_gloo.log; exit 1; }
tail -1 $O/bench_2ranks_gloo.log | cut -c1-2500
timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1 || { tail -30 $O/shard_iter.log; exit 1; }
grep -v amdgpu $O/shard_iter.log
L=tools/analysis
V="base=espm_amd/lib/libespm_mu.so pro=$L/libespm_mu_pro.so red1=$L/libespm_mu_red1.so prio8=$L/libespm_mu_prio8.so s3=$L/libespm_mu_s3.so"
ROWS=64 FUSED=always timeout -k 10 200 python $L/variant_ab.py $V > $O/ab_64_fused.log 2>&1 || { tail -20 $O/ab_64_fused.log; exit 1; }
grep "best\|rel dloss" $O/ab_64_fused.log
ROWS=64 FUSED=0 timeout -k 10 200 python $L/variant_ab.py base=espm_amd/lib/libespm_mu.so > $O/ab_64_two.log 2>&1 || { tail -20 $O/ab_64_two.log; exit 1; }
grep "best" $O/ab_64_two.log
ROWS=128 FUSED=always timeout -k 10 200 python $L/variant_ab.py $V > $O/ab_128_fused.log 2>&1 || { tail -20 $O/ab_128_fused.log; exit 1; }
grep "best" $O/ab_128_fused.log
timeout -k 10 300 python $L/variant_ab.py base=espm_amd/lib/libespm_mu.so pro=$L/libespm_mu_pro.so red1=$L/libespm_mu_red1.so s3=$L/libespm_mu_s3.so > $O/ab_512.log 2>&1 || { tail -20 $O/ab_512.log; exit 1; }
grep "best\|rel dloss" $O/ab_512.log
