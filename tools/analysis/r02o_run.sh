set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
for i in 1 2; do
ESPM_MU_LIB=$R/tools/analysis/libespm_mu_fullpf2.so python bench.py --no-cpu --no-autotune 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf2', d['value'], d.get('steady_state'))"
python bench.py --no-cpu --no-autotune 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf1', d['value'], d.get('steady_state'))"
done
