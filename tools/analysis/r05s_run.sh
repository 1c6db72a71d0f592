O=gpurun_out/r05s; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so k5b128=tools/analysis/libespm_mu_k5b128.so k5b64=tools/analysis/libespm_mu_k5b64.so"
ROWS=512 K=5 REPS=6 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_512.log 2>&1; tail -1 $O/ab_k5_512.log
ROWS=64 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_64.log 2>&1; tail -1 $O/ab_k5_64.log
ROWS=128 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_128.log 2>&1; tail -1 $O/ab_k5_128.log
