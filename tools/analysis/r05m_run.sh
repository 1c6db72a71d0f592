O=gpurun_out/r05m; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so"
for n in skew4 skew8 skew12 skew16; do V="$V $n=tools/analysis/libespm_mu_$n.so"; done
ROWS=64 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_64.log 2>&1; tail -1 $O/ab_k5_64.log
ROWS=128 K=5 ITERS=1000 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_128.log 2>&1; tail -1 $O/ab_k5_128.log
ROWS=256 K=5 ITERS=500 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_k5_256.log 2>&1; tail -1 $O/ab_k5_256.log
CONFIG=c5 ROWS=128 ITERS=500 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_c5_128.log 2>&1; tail -1 $O/ab_c5_128.log
