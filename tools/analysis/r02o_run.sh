set -e
R=$GRAFT_REPO_ROOT
cd $R
for k in 7 8; do
ESPM_MU_LIB=$R/tools/analysis/libespm_mu_wfk8.so K=$k python tools/analysis/default_args_iter.py 2>&1 | grep "reference default" | sed "s/^/k=$k 512 threads: /"
K=$k python tools/analysis/default_args_iter.py 2>&1 | grep "reference default" | sed "s/^/k=$k 1024 threads: /"
done
