O=gpurun_out/r05k; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so"
for n in prio0 prio8 prio20 pf2 unrh6 unrw6 unr3 cutb; do V="$V $n=tools/analysis/libespm_mu_$n.so"; done
ROWS=512 K=5 timeout -k 10 500 python tools/analysis/variant_ab.py $V > $O/ab_k5_512.log 2>&1; tail -1 $O/ab_k5_512.log
ROWS=512 K=8 timeout -k 10 500 python tools/analysis/variant_ab.py $V > $O/ab_k8_512.log 2>&1; tail -1 $O/ab_k8_512.log
ROWS=512 K=3 timeout -k 10 500 python tools/analysis/variant_ab.py $V > $O/ab_k3_512.log 2>&1; tail -1 $O/ab_k3_512.log
CONFIG=c5 ROWS=1024 ITERS=100 REPS=3 timeout -k 10 900 python tools/analysis/variant_ab.py base=espm_amd/lib/libespm_mu.so prio0=tools/analysis/libespm_mu_prio0.so prio20=tools/analysis/libespm_mu_prio20.so unrh6=tools/analysis/libespm_mu_unrh6.so unr3=tools/analysis/libespm_mu_unr3.so > $O/ab_c5_1024.log 2>&1; tail -1 $O/ab_c5_1024.log
