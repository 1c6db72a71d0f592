O=gpurun_out/r05t; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so pp=tools/analysis/libespm_mu_pp.so"
ROWS=512 K=5 REPS=6 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_512.log 2>&1; tail -1 $O/ab_k5_512.log
ROWS=512 K=8 REPS=4 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k8_512.log 2>&1; tail -1 $O/ab_k8_512.log
ROWS=512 K=3 REPS=4 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k3_512.log 2>&1; tail -1 $O/ab_k3_512.log
ROWS=64 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_64.log 2>&1; tail -1 $O/ab_k5_64.log
CONFIG=c5 ROWS=128 ITERS=500 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_c5_128.log 2>&1; tail -1 $O/ab_c5_128.log
