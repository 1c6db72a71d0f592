set -e
O=gpurun_out/r03a; mkdir -p $O
L=tools/analysis
V="base=espm_amd/lib/libespm_mu.so klprod=$L/libespm_mu_klprod.so plain=$L/libespm_mu_plain.so perm=$L/libespm_mu_perm.so all3=$L/libespm_mu_all3.so"
timeout -k 10 300 python $L/variant_ab.py $V > $O/ab_512.log 2>&1 || { tail -20 $O/ab_512.log; exit 1; }
grep -v amdgpu.ids $O/ab_512.log | tail -12
ROWS=64 FUSED=0 timeout -k 10 200 python $L/variant_ab.py $V > $O/ab_64.log 2>&1 || { tail -20 $O/ab_64.log; exit 1; }
grep "best\|rel dloss" $O/ab_64.log
ROWS=128 FUSED=0 timeout -k 10 200 python $L/variant_ab.py $V > $O/ab_128.log 2>&1 || { tail -20 $O/ab_128.log; exit 1; }
grep "best" $O/ab_128.log
timeout -k 10 500 python -m pytest tests/test_gpu_updates.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
