set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || (tail -40 $O/pytest.log | cut -c1-300; exit 1)
tail -1 $O/pytest.log
ROWS=128 python tools/analysis/c5_iter.py 2>&1 | tail -1
python tools/analysis/c5_iter.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksc -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
for f in $O/ksc/*/*_kernel_stats.csv; do (head -1 $f; grep "espm::" $f) > $O/ks_c5_128rows_kernel_stats.csv; done
rm -rf $O/ksc
python3 - <<'PY'
import csv,os
for row in csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r02p/ks_c5_128rows_kernel_stats.csv')):
    if any(t in row['Name'] for t in ('fused','w_reduce','w_finish')):
        print(row['Name'][:70], row['Calls'], round(float(row['AverageNs'])/1e3,1), round(float(row['MinNs'])/1e3,1))
PY
