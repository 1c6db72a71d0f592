# scratch script of the last verification run of round 2 (GPU box): C5 kernel stats with the final code
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python3 $R/tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1
for f in $O/ks_c5/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks_c5_kernel_stats.csv; done
rm -rf $O/ks_c5
grep "C5 rows" $O/c5_iter.log
grep "fused\|w_finish\|w_reduce" $O/ks_c5_kernel_stats.csv | cut -d, -f1-4,6 | cut -c1-200
