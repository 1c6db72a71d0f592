# Counters of the 17..32-component build's dense kernels at the headline image (8-bit and bf16 stores, k = 17 and 32): separate --pmc passes,
# --kernel-trace only.  TAG=r05z bash tools/analysis/prof_wide32_pmc.sh (on the GPU box)
set -e
TAG=${TAG:-r05z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG/w32pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export K="17 32"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq1 -- python3 $R/tools/analysis/wide_iter.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/analysis/wide_iter.py > /dev/null 2>&1 || true
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python3 $R/tools/analysis/wide_iter.py > /dev/null 2>&1
done
cd $R
for d in sq1 sq2 FETCH_SIZE WRITE_SIZE; do
  for f in $O/$d/*/*_counter_collection.csv; do [ -f "$f" ] && cp "$f" $R/gpurun_out/$TAG/w32pmc_${d}_counter_collection.csv; done
done
ls -la $R/gpurun_out/$TAG/ | grep w32pmc
