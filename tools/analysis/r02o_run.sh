set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
python tools/analysis/default_args_iter.py 2>&1 | grep -v amdgpu.ids | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksd -- python3 $R/tools/analysis/default_args_iter.py > /dev/null 2>&1
for f in $O/ksd/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks_default_args_kernel_stats.csv; done
rm -rf $O/ksd
cut -d, -f1-7 $O/ks_default_args_kernel_stats.csv | head -8 | cut -c1-220
