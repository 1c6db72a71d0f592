# Configuration 5's fused kernel (mu_fused_ell_kernel<8, ...>) under rocprofv3: kernel stats and the four PMC passes the headline has
# (FETCH_SIZE, WRITE_SIZE, two SQ sets; separate passes, --kernel-trace only).  TAG=r05a bash tools/analysis/prof_c5_pmc.sh
set -e
TAG=${TAG:-r05a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R && python tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python3 $R/tools/analysis/c5_iter.py > $O/c5_iter_ks.log 2>&1
export ITERS=30
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c5_$C -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
done
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_c5_sq1 -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c5_sq2 -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
cd $R
for d in ks_c5 pmc_c5_FETCH_SIZE pmc_c5_WRITE_SIZE pmc_c5_sq1 pmc_c5_sq2; do
  for f in $O/$d/*/*_kernel_stats.csv $O/$d/*/*_counter_collection.csv; do
    [ -f "$f" ] || continue
    (head -1 $f; grep "espm::" $f) > $O/${d}_$(basename $f | sed 's/^[0-9]*_//')
  done
  rm -rf $O/$d
done
ls -la $O
cat $O/c5_iter.log
