O=gpurun_out/r05o; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so sum8=tools/analysis/libespm_mu_sum8.so"
ROWS=512 K=5 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_512.log 2>&1; tail -1 $O/ab_k5_512.log
ROWS=512 K=8 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k8_512.log 2>&1; tail -1 $O/ab_k8_512.log
ROWS=64 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_64.log 2>&1; tail -1 $O/ab_k5_64.log
ROWS=128 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_128.log 2>&1; tail -1 $O/ab_k5_128.log
CONFIG=c5 ROWS=128 ITERS=500 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_c5_128.log 2>&1; tail -1 $O/ab_c5_128.log
CONFIG=c5 ROWS=1024 ITERS=100 REPS=3 timeout -k 10 900 python tools/analysis/variant_ab.py $V > $O/ab_c5_1024.log 2>&1; tail -1 $O/ab_c5_1024.log
