set -e
O=gpurun_out/r03i; mkdir -p $O
L=tools/analysis
V="base=espm_amd/lib/libespm_mu.so pre1=$L/libespm_mu_pre1.so"
for R in 64 128 256; do
  ROWS=$R timeout -k 10 200 python $L/variant_ab.py $V > $O/ab_$R.log 2>&1 || { tail -20 $O/ab_$R.log; exit 1; }
  echo "rows $R"; grep "best\|rel dloss" $O/ab_$R.log
done
HSA_ENABLE_SDMA=0 timeout -k 10 300 python $L/fit_timing.py > $O/fit_timing_nosdma.log 2>&1 || { tail -30 $O/fit_timing_nosdma.log; exit 1; }
grep -v amdgpu $O/fit_timing_nosdma.log
