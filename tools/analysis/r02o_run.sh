set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -40 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksd -- python3 $R/tools/analysis/default_args_iter.py > $O/default_args_iter.log 2>&1
for f in $O/ksd/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks_default_args_kernel_stats.csv; done
rm -rf $O/ksd
grep "us/iteration" $O/default_args_iter.log
grep "w_finish_fast" $O/ks_default_args_kernel_stats.csv | cut -d, -f1-7 | cut -c1-200
