O=gpurun_out/r05w; mkdir -p $O
V="base=espm_amd/lib/libespm_mu.so half6=tools/analysis/libespm_mu_half6.so half6pp=tools/analysis/libespm_mu_half6pp.so"
for KK in 6 7 8; do ROWS=512 K=$KK REPS=4 timeout -k 10 300 python tools/analysis/variant_ab.py $V > $O/ab_k${KK}_512.log 2>&1; echo "k $KK: $(tail -1 $O/ab_k${KK}_512.log)"; done
CONFIG=c5 ROWS=128 ITERS=500 timeout -k 10 400 python tools/analysis/variant_ab.py $V > $O/ab_c5_128.log 2>&1; echo "c5 shard: $(tail -1 $O/ab_c5_128.log)"
CONFIG=c5 ROWS=1024 ITERS=100 REPS=3 timeout -k 10 900 python tools/analysis/variant_ab.py $V > $O/ab_c5_1024.log 2>&1; echo "c5: $(tail -1 $O/ab_c5_1024.log)"
