# Profiles of round 5 (rocprofv3 on the GPU box).  TAG=r05fin PARTS="1 2 3 4" bash tools/analysis/prof_round_r05.sh
# (a gpurun call is at most 20 minutes: PARTS="1" bench + phase stamps + shard iterations, "2" kernel stats / summaries, "3" the headline's
#  and configuration 5's counters, "4" the dense 8-bit store's kernel stats and counters)
#   bench (default and the driver's 20 / 5 protocol); kernel stats AND per-dispatch summaries (median, trimmed mean: tools/kstats.py on the
#   kernel trace) of the bench loop, of configuration 5, of the 64- / 128-row shards and of configuration 5's 128-row shard with the full
#   exchange protocol (group of one rank); HBM traffic and SQ counters of the headline's AND configuration 5's kernels (separate --pmc
#   passes, --kernel-trace only); phase stamps.  "quick": the kernel stats / summaries only.
set -e
TAG=${TAG:-r05fin}
PARTS=${PARTS:-"1 2 3 4"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if has 1; then
  python bench.py > $O/bench_default.log 2> $O/bench_default.err
  python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2> $O/bench_20_5.err
  for RW in 512 128 64; do ROWS=$RW python tools/analysis/phase_clock.py > $O/phase_clock_${RW}rows.log 2>&1 || true; done
  CONFIG=c5 ROWS=1024 python tools/analysis/phase_clock.py > $O/phase_clock_c5_1024.log 2>&1 || true
  CONFIG=c5 ROWS=128 python tools/analysis/phase_clock.py > $O/phase_clock_c5_128.log 2>&1 || true
  for RW in 64 128; do ROWS=$RW python tools/analysis/shard_iter.py > $O/shard_iter_$RW.log 2>&1; done
  CONFIG=c5 ROWS=128 python tools/analysis/shard_iter.py > $O/shard_iter_c5_128.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
if has 2; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --no-cpu --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python3 $R/tools/analysis/c5_iter.py > $O/c5_iter.log 2>&1
ROWS=64 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard64 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard128 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
CONFIG=c5 ROWS=128 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_shard_c5_128 -- python3 $R/tools/analysis/shard_iter.py > /dev/null 2>&1
fi
if has 3; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --no-cpu --no-extras --no-autotune --steps 20 --warmup 5 > /dev/null 2>&1
  export ITERS=30
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c5_$C -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_c5_sq1 -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c5_sq2 -- python3 $R/tools/analysis/c5_iter.py > /dev/null 2>&1
  unset ITERS
fi
if has 4; then
  # the dense 8-bit store's kernels (the data-independent path's counters had been round 1's)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_u8 -- python3 $R/bench.py --no-cpu --no-extras --x-store u8 --steps 100 --warmup 10 > $O/bench_u8.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_u8_$C -- python3 $R/bench.py --no-cpu --no-extras --x-store u8 --steps 20 --warmup 5 > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_u8_sq1 -- python3 $R/bench.py --no-cpu --no-extras --x-store u8 --steps 20 --warmup 5 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u8_sq2 -- python3 $R/bench.py --no-cpu --no-extras --x-store u8 --steps 20 --warmup 5 > /dev/null 2>&1
fi
cd $R
for d in ks ks_c5 ks_shard64 ks_shard128 ks_shard_c5_128 ks_u8 pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq1 pmc_sq2 pmc_c5_FETCH_SIZE pmc_c5_WRITE_SIZE pmc_c5_sq1 pmc_c5_sq2 pmc_u8_FETCH_SIZE pmc_u8_WRITE_SIZE pmc_u8_sq1 pmc_u8_sq2; do
  [ -d $O/$d ] || continue
  for f in $O/$d/*/*_kernel_stats.csv $O/$d/*/*_counter_collection.csv; do
    [ -f "$f" ] || continue
    (head -1 $f; grep "espm::" $f) > $O/${d}_$(basename $f | sed 's/^[0-9]*_//')
  done
  case $d in ks*)
    for f in $O/$d/*/*_kernel_trace.csv; do
      [ -f "$f" ] || continue
      python3 tools/kstats.py $f --csv $O/${d}_kernel_summary.csv > $O/${d}_kernel_summary.txt
    done;;
  esac
  rm -rf $O/$d
done
ls -la $O
[ -f $O/bench_default.log ] && tail -1 $O/bench_default.log | cut -c1-300
