set -e
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
ESPM_MU_LIB=$R/tools/analysis/libespm_mu_wf256.so python tools/analysis/default_args_iter.py 2>&1 | grep "reference default" | sed 's/^/256: /'
python tools/analysis/default_args_iter.py 2>&1 | grep "reference default" | sed 's/^/512: /'
done
