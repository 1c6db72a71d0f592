O=gpurun_out/r05u; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so pp7=tools/analysis/libespm_mu_pp7.so"
for KK in 1 2 3 4 5 6 7; do ROWS=512 K=$KK REPS=4 timeout -k 10 300 python tools/analysis/variant_ab.py $V > $O/ab_k${KK}_512.log 2>&1; echo "k $KK: $(tail -1 $O/ab_k${KK}_512.log)"; done
COUNTS=100 ROWS=512 K=5 REPS=4 timeout -k 10 300 python tools/analysis/variant_ab.py $V > $O/ab_k5_512_counts100.log 2>&1; echo "k 5, 100 counts: $(tail -1 $O/ab_k5_512_counts100.log)"
COUNTS=2000 ROWS=512 K=5 REPS=4 timeout -k 10 300 python tools/analysis/variant_ab.py $V > $O/ab_k5_512_counts2000.log 2>&1; echo "k 5, 2000 counts: $(tail -1 $O/ab_k5_512_counts2000.log)"
