# Profiles of round 4 (rocprofv3 on the GPU box): bench (default and the driver's 20 / 5 protocol), kernel stats of the bench loop, HBM
# traffic and SQ counters of its kernels (separate --pmc passes, --kernel-trace only), kernel stats of C5, of the 64- / 128-row shards
# with the full exchange protocol (group of one rank) and of configuration 5's 128-row shard; phase stamps.  TAG=r04fin bash tools/analysis/prof_round_r04.sh
set -e
TAG=${TAG:-r04fin}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R && python bench.py > $O/bench_default.log 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2> $O/bench_20_5.err
for RW in 512 128 64; do ROWS=$RW python tools/analysis/phase_clock.py > $O/phase_clock_${RW}rows.log 2>&1; done
for RW in 64 128; do ROWS=$RW python tools/analysis/shard_iter.py > $O/shard_iter_$RW.log 2>&1; done
CONFIG=c5 ROWS=128 python tools/analysis/shard_iter.py > $O/shard_iter_c5_128.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --no-cpu --no-extras > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python3 $R/tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1
ROWS=64 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard64 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard128 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
CONFIG=c5 ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard_c5_128 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
cd $R
for d in ks ks_c5 ks_shard64 ks_shard128 ks_shard_c5_128 pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq1 pmc_sq2; do
  for f in $O/$d/*/*_kernel_stats.csv $O/$d/*/*_counter_collection.csv; do
    [ -f "$f" ] || continue
    (head -1 $f; grep "espm::" $f) > $O/${d}_$(basename $f | sed 's/^[0-9]*_//')
  done
  rm -rf $O/$d
done
ls -la $O
tail -1 $O/bench_default.log | cut -c1-300
