set -e
O=gpurun_out/r03l; mkdir -p $O
L=tools/analysis
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_sharded_fullsize.py tests/test_gpu_estimator.py -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for R in 1024 128; do
  for S in 1 0; do
    ROWS=$R ESPM_W_GSPLIT=$S timeout -k 10 200 python $L/c5_iter.py 2>&1 | grep "C5" | sed "s/^/gsplit=$S /" | tee -a $O/c5_iter.log
  done
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
cd $R
for f in $O/ks/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks_c5_128rows_kernel_stats.csv; done
rm -rf $O/ks
python tools/kstats.py $O/ks_c5_128rows_kernel_stats.csv | head -8
