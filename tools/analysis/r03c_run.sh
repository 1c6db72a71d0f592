set -e
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
ESPM_DEBUG_RESIDENCY=1 timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter.log 2>&1 || { tail -30 $O/shard_iter.log; exit 1; }
grep -v "amdgpu\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" $O/shard_iter.log
ESPM_FUSED=always timeout -k 10 200 python tools/analysis/shard_iter.py > $O/shard_iter_fused.log 2>&1 || { tail -30 $O/shard_iter_fused.log; exit 1; }
grep "us/it" $O/shard_iter_fused.log
for R in 64 128 512; do
  ROWS=$R FUSED=always timeout -k 10 200 python tools/analysis/phase_clock.py > $O/phase_clock_${R}rows.log 2>&1 || { tail -30 $O/phase_clock_${R}rows.log; exit 1; }
  grep -v amdgpu $O/phase_clock_${R}rows.log
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
ESPM_FUSED=always rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks64 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
cd $R
for f in $O/ks64/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks64_fused_kernel_stats.csv; done
rm -rf $O/ks64
python tools/kstats.py $O/ks64_fused_kernel_stats.csv
