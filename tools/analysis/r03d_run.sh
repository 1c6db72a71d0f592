set -e
O=gpurun_out/r03d; mkdir -p $O
L=tools/analysis
for v in "" "$PWD/$L/libespm_mu_wide_k32pad.so" "$PWD/$L/libespm_mu_wide_k16.so"; do
  ESPM_MU_WIDE_LIB=${v:-$PWD/espm_amd/lib/libespm_mu_wide.so} timeout -k 10 200 python $L/wide_repro.py 2>&1 | grep -v amdgpu | tee -a $O/wide_repro.log
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect "tests/test_gpu_fullsize.py::test_matrix_core_kernels_of_the_wide_build_at_full_size" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1 || { tail -30 $O/shard_iter.log; exit 1; }
grep "us/it\|lost" $O/shard_iter.log
ESPM_FUSED=always timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_fused.log 2>&1 || { tail -30 $O/shard_iter_fused.log; exit 1; }
grep "us/it" $O/shard_iter_fused.log
ROWS=128 ESPM_FUSED=always timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_fused_128.log 2>&1 || { tail -30 $O/shard_iter_fused_128.log; exit 1; }
grep "us/it" $O/shard_iter_fused_128.log
ROWS=256 timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_256.log 2>&1 || { tail -30 $O/shard_iter_256.log; exit 1; }
grep "us/it" $O/shard_iter_256.log
for R in 64; do
  ROWS=$R FUSED=always timeout -k 10 200 python tools/analysis/phase_clock.py > $O/phase_clock_${R}rows.log 2>&1 || { tail -30 $O/phase_clock_${R}rows.log; exit 1; }
  grep -v amdgpu $O/phase_clock_${R}rows.log
done
timeout -k 10 300 python tools/analysis/fused_check.py > $O/fused_check.log 2>&1 || { tail -30 $O/fused_check.log; exit 1; }
grep "fused=\|max |dW\|FUSED_OK" $O/fused_check.log | tail -8
