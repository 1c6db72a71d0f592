// One launch for both half-steps of an iteration on the sparse count store (x_dtype = ESPM_X_ELL).
//
// The W accumulation of a pixel block needs the new H of THAT block only (updates.py:38-39, :53-59: R = X / (GW H') and
// R H'^T are sums over pixels), so the workgroup that has just updated the 1024 pixels of a block can go on and walk the
// block's channel lists without waiting for anybody else: no grid-wide dependency sits between the two half-steps.
// What IS global - the row sums of H' in the denominator of W', the sum of the slabs - stays in the reduction launch
// that follows (w_reduce_update_kernel / w_reduce_kernel, mu_w_step.hip).
//
//   workgroup = 16 waves = one block of ESPM_ELL_PB = 1024 pixels = two H tiles of 512 pixels (16 pixel-list groups);
//   images that do not fill the chip with such blocks (and the shards of a sharded image) have blocks of ell_pb = 128 .. 512
//   pixels = two H tiles of 64 .. 256, about one per compute unit, and a workgroup of 8 waves (NT = 512; the block
//   size is then a run-time value)
//   1. GW table -> LDS (address 0), as in h_step_ell_kernel
//   2. H walk, DYNAMIC: the rows of every list group are cut into S segments; the 16 S (group, segment) units are handed
//      out through a counter in LDS, so a wave that is done takes the next unit.  A unit's partial numerators (and its
//      part of the KL sum) go to slot `segment` of its pixels, and a pixel's slots are summed in slot order: the result
//      does not depend on which wave walked what (bit-reproducible), only the time does.
//      Why: the four waves of a SIMD do not share it evenly - with equal shares of the rows, waves 0-3 of the workgroup
//      finished their walk after 42 us, waves 12-15 after 71 (tools/analysis/phase_clock.py), and the tail ran one wave
//      per SIMD deep.
//   3. epilogue, one thread per pixel (h_epilogue): H' -> h[1 - src] (the next iteration's stencil and rel_H need it in
//      memory) and, instead of the transposed copy h_t in memory, straight into the LDS table of the W walk, which
//      takes the place of the GW table (address 0: the unit entries of both list sets address their table without a base)
//   4. W walk, dynamic too: the block's channel groups (in order of decreasing length) through a second counter; a group
//      is walked by one wave, so its rows of the slab have one writer whoever that is.
//
// Against the two launches this saves a kernel boundary, the table prologue of the W accumulation (1024 rows of h_t
// from memory, a barrier), and 2 x 4 KP p bytes of h_t traffic.  The block's record goes to the hpart slot of its first
// tile, zeros to the slot of the second (HStepArgs::rec_nb), so the readers of the records do not change.
#pragma once
#include "mu_ell_kernel.hpp"

// the fused kernel below the full geometry (blocks of 128 .. 512 pixels): threads of a workgroup, list batches in flight
#ifndef ESPM_FUSED_SMALL_THREADS
#define ESPM_FUSED_SMALL_THREADS 1024
#endif
#ifndef ESPM_FUSED_SMALL_PREFETCH
#define ESPM_FUSED_SMALL_PREFETCH 1
#endif
// first segment of a list group split in two (k = 7, 8), per cent of its rows
#ifndef ESPM_FUSED_CUT2
#define ESPM_FUSED_CUT2 60
#endif
// Below the full geometry a wave's FIRST unit of the W walk is its own (channel group = wave number; the counter hands out the rest) and
// what the unit starts with - the group's channels and list offsets (LDS), its rows of G W and first list rows (memory) - is requested
// behind the wave's pixels' update, ahead of the record reduction and its barrier: all 16 waves start the walk together, so nobody
// covered that round trip.  64 rows 32.2 -> 31.7 us, 128 rows 44.6 -> 43.6, 256 rows 71.6 -> 70.0 (profiles/r04ax_*); at the full
// geometry (4 units per wave) nothing - left out there.  The same for the H walk's first unit (offsets, slot -> pixel, H column, first
// rows requested while the table is staged) was SLOWER at every size (+1.5 ... +6 us, profiles/r04aw_*): its two levels of dependent
// loads sit in the instruction stream of a wave that issues in order, ahead of or behind the staging loads - withdrawn.
#ifndef ESPM_FUSED_FIRST_PREFETCH
#define ESPM_FUSED_FIRST_PREFETCH 1
#endif
#ifndef ESPM_FUSED_FULL_PREFETCH
#define ESPM_FUSED_FULL_PREFETCH 1
#endif
// Round 3, measured together (tools/analysis/variant_ab.py, profiles/r03b_variant_ab_*.log; same results bit for bit): a 64-row shard
// of the headline image 46.3 -> 43.5 us per iteration with the fused launch, a 128-row one 57.2 -> 54.4, the headline 147.4 -> 145.9.
// the record reduction with ONE barrier (h_epilogue, red_scratch)
#ifndef ESPM_FUSED_RED_ONE_BARRIER
#define ESPM_FUSED_RED_ONE_BARRIER 1
#endif
// below the full geometry: a wave's issue priority follows the rows it still has to walk, in steps of this many dwords (0: off)
// (round 4: 4 dwords - only a unit's last dozen rows yield - instead of round 3's 8, which was the worst of 2 ... 24: 128 rows 41.4 ->
//  40.6 us, 256 rows 67.7 -> 66.8, 64 rows 30.9 -> 30.7, profiles/r04bd_*, r04be_*; 16 is better still on 128-pixel blocks (30.4) and worse
//  on configuration 5's; a step chosen at run time by the block size lost what it gained to its scalar compares, r04bf_*)
#ifndef ESPM_FUSED_SMALL_PRIO
#define ESPM_FUSED_SMALL_PRIO 4
#endif
// The same at the full geometry (four units per wave and walk, handed out dynamically), from 5 components on - where a table row is more
// than one 16-byte gather: step 12: headline (k = 5) 131.8 -> 130.2 us, k = 6 146.8 -> 140.7, configuration 5 (k = 8) 648 -> 641; k = 2 ... 4
// lose 0.1 ... 1 us with any step and stay without (profiles/r04bh_*, r04bi_*).  0 = off.
#ifndef ESPM_FUSED_FULL_PRIO
#define ESPM_FUSED_FULL_PRIO 12
#endif
// the prologue's global loads issued together (the kernel's comment at its prologue)
#ifndef ESPM_FUSED_PROLOGUE_BATCH
#define ESPM_FUSED_PROLOGUE_BATCH 1
#endif

// the per-pixel sum of the partial numerators with all its LDS reads in flight together (h_epilogue, MAXP)
#ifndef ESPM_FUSED_SUM_BATCHED
#define ESPM_FUSED_SUM_BATCHED 1
#endif
// the block's slab of R H'^T collected in LDS (the numerators' region, free from the epilogue's barrier on) and written out as whole
// rows in 16-byte write-through stores, instead of 4-byte stores scattered by the channel order: 2.6 M scattered dword stores per
// launch become 0.66 M coalesced 16-byte ones, and the launch ends without 10.5 MB of dirty lines for the boundary to write back
// (MI355X_MICROARCH.md: + B / 6 TB/s behind B dirty bytes; write-through wins for tens of KB per workgroup)
#ifndef ESPM_FUSED_SLAB_LDS
#define ESPM_FUSED_SLAB_LDS 1
#endif
// half channel groups as the units of the W walk at the full geometry (FusedArgs::w_split).  Measured (round 4, profiles/r04e_w_split_ab_*.log): SLOWER -
// 140.5 against 138.1 us per iteration at the headline, 118.3 against 115.2 at k = 3, 48.5 against 47.0 at 100 counts per pixel: 64 unit
// starts per workgroup (GW rows, first list rows: a memory round trip each) cost more than the shorter tail returns.  Off.
// most segments per list group below the full geometry (FusedArgs::h_segs; the per-pixel sums unroll to it: at 16 the compiler keeps 96
// partial values in flight and spills 224 registers).  MORE segments than 1024 / pb - units left over to level the waves - were measured
// where the region grown for the slab holds them (ESPM_FUSED_SMALL_SEGS): slower everywhere - 64 rows 48.7 -> 50.4 us per iteration at
// 10-13 segments, 128 rows 62.9 -> 65.5-66.2 at 5-6, 256 rows 96.9 -> 101.4 at 3, configuration 5's 128-row shard 144.1 -> 149.4 at 3
// (all on the 16-wide build, itself 13 us slower than the 8-wide one: profiles/r04m_small_segs_ab_*.log).  A unit's start costs more
// than its levelling returns, as in the W walk's half units.
#ifndef ESPM_FUSED_MAX_SEGS
#define ESPM_FUSED_MAX_SEGS 8
#endif
// Below the full geometry the H walk has as many units as the workgroup has waves (16): one each, all of the same length - and the waves of a
// SIMD are served oldest first, so waves 12-15 of configuration 5's 128-row shard end their unit 5 us after waves 0-3 (35.2 against 29.8 us,
// profiles/r05fin_phase_clock_c5_128.log) and the walk ends with the SIMDs one wave deep.  ESPM_FUSED_SMALL_SKEW (per cent): wave w walks unit w
// (segment w / groups: the older waves of every SIMD hold the earlier segments) and the segments shrink linearly from (100 + skew) % to
// (100 - skew) % of the even share, so that the younger waves have less to walk.  The partial numerators are summed by segment index as before:
// results do not depend on it beyond the order of the rows inside a pixel's sum, which follows the cuts.  0: even segments, dynamic hand-out.
// MEASURED (profiles/r05m_ab_*.log, skew 0 | 4 | 8 | 12 | 16, us per iteration): 64 rows 31.0 | 30.9 | 30.8 | 30.7 | 30.8, 128 rows 41.8 | 42.3 | 42.3 | 42.2 | 42.2,
// 256 rows 68.2 | 70.5 | 69.3 | 69.1 | 68.9, configuration 5's shard 93.9 | 94.2 | 94.8 | 94.3 | 94.4: nothing - the fixed assignment loses what the shorter units of the
// young waves gain (with the dynamic hand-out a wave that starts late simply takes a later unit).  Off.
#ifndef ESPM_FUSED_SMALL_SKEW
#define ESPM_FUSED_SMALL_SKEW 0
#endif
#ifndef ESPM_FUSED_W_SPLIT
#define ESPM_FUSED_W_SPLIT 0
#endif
#ifndef ESPM_FUSED_SLAB_WT   // the rows written through (sc0 sc1) or as plain stores (A/B)
#define ESPM_FUSED_SLAB_WT 1
#endif

namespace espm {

struct FusedArgs {
  HStepArgs h;      // write_h = 1, ell_tp = w.pb / 2; h_t unused
  WAccumArgs w;     // h_t unused
  int cnt_lds_off;  // byte offset of the two unit counters in LDS; behind them (+ 16) the H walk's units leave their KL sums: ESPM_FUSED_MAX_UNITS floats
  int meta_lds_off; // byte offset of the block's list offsets in LDS: pb / 64 x (first row, first general row, end) of the
                    // pixel-list groups, then the 2 n_cg + 1 offsets of the block's channel-list groups
  int perm_lds_off; // below the full geometry: byte offset of the block's copy of pix_perm (pb ints) and chan_perm (64 n_cg ints) in LDS
  int static_units; // A/B only (espm_mu_state.no_fused = 2): wave w takes the units w, w + 16, ... instead of the next free one
  int red_lds_off;  // byte offset of the scratch of the record reduction (16 waves x 21 doubles), < 0: the numerators' region after a barrier of its own
  int perm_lds;     // the block's pix_perm / chan_perm are copied to LDS (always below the full geometry; at the full geometry where they fit)
  int h_segs;       // segments per list group of the H walk: below the full geometry at least 1024 / pb, at most ESPM_FUSED_MAX_SEGS; at the full
                    // geometry from 6 components on as many as the LDS holds partials for (FusedGeom<K>::S_MIN .. S_MAX)
  int w_split;      // the W walk's units are half channel groups, summed through two copies of the slab in LDS (needs slab_lds and room for the second copy)
  int stream_lists; // espm_mu_state.ell_stream: the STREAM instance where one is built (the full geometry's lean instances)
  int slab_lds;     // the W walk collects the block's slab in the numerators' region and the workgroup writes it out as rows (the launcher: where k n_pad floats fit there and the record reduction has scratch of its own)
};

// (the cuts of four segments, cumulative per cent of a group's rows: A/B knobs - profiles/r03bf_*)
#ifndef ESPM_FUSED_CUT4_A
#define ESPM_FUSED_CUT4_A 45
#endif
#ifndef ESPM_FUSED_CUT4_B
#define ESPM_FUSED_CUT4_B 75
#endif
#ifndef ESPM_FUSED_CUT4_C
#define ESPM_FUSED_CUT4_C 92
#endif
// Segments per list group of the H walk: as many as the LDS holds partials for (K rows of pb floats each).
// Round 5: a unit's part of the KL sum no longer occupies a row of the partials - the unit's wave adds its 64 lanes' values and leaves ONE float
// (FusedArgs::cnt_lds_off + 16 + 4 unit; summed per unit index, not per walker: the loss stays bit-reproducible) - so a third segment fits at
// k = 7, 8 where the table leaves room (configuration 5: 1984 rows of 32 bytes + 3 x 8 x 4 KB, with the record reduction's scratch in the part
// of the table that is dead once the H walk is over) and a fourth at k = 6: the launcher picks S_MIN .. S_MAX (FusedArgs::h_segs).
// Why: at k = 7, 8 two segments were 32 units for 16 waves - every wave one long and one short unit, nothing to hand to a wave that is done
// early: the slowest wave ended 7.8 us after wave 0 of a 65 us walk (profiles/r04ah_phase_c5_1024.log).
#define ESPM_FUSED_MAX_UNITS 64
template <int K>
struct FusedGeom {
  static constexpr int S = K <= 4 ? 4 : (K <= 6 ? 3 : 2);       // S_MIN: what always fits where the fused kernel applies (fused_ell_lds_bytes; from 5 components on a row of the table is 32 bytes: FixTab)
  static constexpr int S_MAX = K <= 6 ? 4 : 3;
  static constexpr int PROWS = K;
  // first row of segment s of a group of `len` rows: the segments shrink (45 / 30 / 17 / 8 % of the rows; 35 / 30 / 20 / 15 until the end
  // of round 3: with the last units half as long the waves of a walk end 1.5 us closer together, profiles/r03bf_*), so that the
  // units handed out last are the short ones and the waves end close together
  // A group of fewer than MIN_SPLIT rows (low doses) is ONE unit: starting a unit costs a chain of dependent loads.
  static constexpr int MIN_SPLIT = 48;
  static __device__ __forceinline__ int seg_begin(int len, int s) {
    if (len < MIN_SPLIT) return s == 0 ? 0 : len;
    constexpr int cut4[5] = {0, ESPM_FUSED_CUT4_A, ESPM_FUSED_CUT4_B, ESPM_FUSED_CUT4_C, 100}, cut3[4] = {0, 45, 80, 100}, cut2[3] = {0, ESPM_FUSED_CUT2, 100};
    const int c = S == 4 ? cut4[s] : (S == 3 ? cut3[s] : cut2[s]);
    return (int)((long)len * c / 100);
  }
  // the full geometry with a run-time number of segments (k >= 6)
  static __device__ __forceinline__ int seg_begin_rt(int len, int s, int segs) {
    if (len < MIN_SPLIT) return s == 0 ? 0 : len;
    constexpr int cut4[5] = {0, ESPM_FUSED_CUT4_A, ESPM_FUSED_CUT4_B, ESPM_FUSED_CUT4_C, 100}, cut3[4] = {0, 45, 80, 100}, cut2[3] = {0, ESPM_FUSED_CUT2, 100};
    const int c = segs == 4 ? cut4[s] : (segs == 3 ? cut3[s] : cut2[s]);
    return (int)((long)len * c / 100);
  }
  // below the full geometry: `segs` equal segments
  static __device__ __forceinline__ int seg_begin_even(int len, int s, int segs) {
    if (len < MIN_SPLIT) return s == 0 ? 0 : len;
    if constexpr (ESPM_FUSED_SMALL_SKEW != 0) {
      // segment j holds (1 + a (1 - (2 j + 1) / segs)) / segs of the rows, a = skew / 100: the first s of them hold s / segs (1 + a (1 - s / segs))
      if (segs > 1) return (int)(((long)len * s * (100L * segs + (long)ESPM_FUSED_SMALL_SKEW * (segs - s))) / (100L * segs * segs));
    }
    return (int)((long)len * s / segs);
  }
};

// PLAIN: the common case as compile-time facts (h_epilogue's PLAIN, plus: the prologue's staged form applies, no extra tail workgroup,
// dynamic units, the slab collected in LDS, scratch of its own for the record reduction) - the launcher checks every one of them on the
// host (launch_fused_k) and takes the generic instance otherwise.  Instantiated in mu_fused_plain.hip.
// STREAM: the lists with non-temporal loads (mu_ell_kernel.hpp: ell_list_load; espm_mu_state.ell_stream) - instances of the full geometry only.
template <int K, bool LOSS, int UNR_H, int UNR_W, int NT, bool FULL, bool PLAIN = false, bool STREAM = false>   // FULL: the full geometry, block and tile sizes are constants
__global__ __launch_bounds__(NT) void mu_fused_ell_kernel(const FusedArgs fa) {
  static_assert(!FULL || NT == ESPM_ELL_WTHREADS, "full geometry: 1024 threads");
  static_assert(FULL || !STREAM, "streamed lists: the full geometry (an image below it fits the last-level cache)");
  static_assert(ESPM_ELL_PB == 2 * ESPM_ELL_TILE && ESPM_ELL_WTHREADS == ESPM_ELL_PB, "full geometry: a workgroup is two H tiles, one thread per pixel");
  const int PB = FULL ? ESPM_ELL_PB : fa.w.pb;     // pixels per workgroup
  const int PBITS = FULL ? ESPM_ELL_PBITS : fa.w.pbits;
  const int TP = PB / 2;                           // pixels per H tile
  const int GPT_SHIFT = PBITS - 7;                 // log2(list groups per tile)
  constexpr int PROWS = FusedGeom<K>::PROWS;
  // segments per list group of the H walk: below the full geometry 1024 / PB, i.e. always 16 (group, segment) units
  constexpr bool S_RT = FULL && FusedGeom<K>::S_MAX > FusedGeom<K>::S;   // (k >= 5 at the full geometry: as many segments as fit, the launcher's choice)
  const int S = (FULL && !S_RT) ? FusedGeom<K>::S : fa.h_segs;   // (below the full geometry: 1024 / PB, or more where the numerators' region - grown for the slab - holds them)
  constexpr int PF = FULL ? ESPM_FUSED_FULL_PREFETCH : ESPM_FUSED_SMALL_PREFETCH;   // list batches requested ahead (ell_walk)
  constexpr bool WALK_PP = FULL && K <= ESPM_ELL_WALK_PP_MAX_K;   // (mu_ell_kernel.hpp: the list dwords in two register sets used in turn)
  constexpr int PRIO = FULL ? (K >= 5 ? ESPM_FUSED_FULL_PRIO : 0) : ESPM_FUSED_SMALL_PRIO;                            // (ell_walk_prio)
  const int NGRP = PB / 64;                        // pixel-list groups of the block
  const HStepArgs& a = fa.h;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;   // [n_pad] rows of GW, later [PB] rows of H'
  ell_table_at_lds_zero(tab);
  ESPM_PHASE_STAMP(0);
  ESPM_PHASE_WHERE();
  const int tab_rows = a.n_pad > PB ? a.n_pad : PB;
  float* part = smem + FixTab<K>::bytes(tab_rows) / sizeof(float);   // (the tables: FixTab, mu_h_kernel.hpp) [S][PROWS][PB] partials (pixel = its place in the block), then reduction scratch
  int* cnt = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(smem) + fa.cnt_lds_off);   // [0]: next unit of the H walk, [1]: of the W walk
  float* ukl = reinterpret_cast<float*>(cnt + 4);   // [NGRP S]: unit u's part of the KL sum
  if (!PLAIN && a.tail_on == 1 && blockIdx.x == gridDim.x - 1) {   // (uniform) the extra workgroup: tail of the previous W update
    w_tail_body<(10 * ESPM_ELL_WTHREADS) / NT>(a.tail, reinterpret_cast<double*>(smem));
    ESPM_PHASE_STAMP(7);   // (instrumented build: when the extra workgroup got a CU - stamp 0 - and when it was done)
    return;
  }
  double* cs_lds = a.cs_parts ? reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(smem) + a.cs_lds_off) : nullptr;
  int* meta = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(smem) + fa.meta_lds_off);
  int* lpix = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(smem) + fa.perm_lds_off);
  int* lchan = lpix + PB;
  const bool perm_lds = PLAIN ? !FULL : (!FULL || fa.perm_lds);   // (uniform)
  // ---- prologue: GW table, the block's list offsets and permutations, the column sums of G W' -> LDS.
  // ESPM_FUSED_PROLOGUE_BATCH: every global load of the prologue is issued before the first result is consumed.  As loops
  // with run-time trip counts (the round-2 form, kept below) each of the five pieces waited for its own loads: five serial
  // round trips to memory, 3.2 us of a workgroup's 127 at the headline and 4.1 of 34 on a 64-row shard.
  // The tail of the W update that produced the input state (tail_on = 2): no extra workgroup - at one workgroup per CU it starts when
  // the first regular one exits and the launch ends with it, 5.5 us after everybody else on a 64-row shard
  // (profiles/r03c_phase_clock_64rows.log).  Instead every workgroup sums the partials it needs anyway (column sums of G W'; with
  // them the sums of W' for its mean), takes the entries of W', W with index in its slice [rw_lo, rw_hi) and leaves its share of
  // rel_W (base.py:323) in its record (ESPM_HP_RELW: the reduction of the records writes the history); workgroup 0 writes colsum_gw.
  const bool spread = a.tail_on == 2 && cs_lds;   // (uniform)
  constexpr int RW = 2;                           // entries of W a thread holds across the prologue
  const int rw_n = a.tail.n * a.tail.k, rw_per = (rw_n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int rw_lo = min(rw_n, (int)blockIdx.x * rw_per), rw_hi = min(rw_n, rw_lo + rw_per);
  const bool rw_staged = rw_per <= RW * NT;
  float rwn[RW], rwo[RW];
#pragma unroll
  for (int u = 0; u < RW; ++u) rwn[u] = rwo[u] = 1.f;
  // (ESPM_FUSED_FIRST_PREFETCH: the W walk's first unit per wave is requested ahead, below)
  constexpr bool OWN_H = PLAIN && !FULL && ESPM_FUSED_SMALL_SKEW != 0;   // (the H walk's first unit per wave is its own: ESPM_FUSED_SMALL_SKEW)
  constexpr bool PREF_W = PLAIN && !FULL && K <= 6 && ESPM_FUSED_FIRST_PREFETCH != 0;   // (k = 7, 8: the request's registers are spilled ones, configuration 5's shard 100.5 -> 101.1 us)
  const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  bool staged = false;
  if constexpr (ESPM_FUSED_PROLOGUE_BATCH) {
    constexpr int TR = 4, PC = 4;   // table rows / permutation entries a thread stages
    const int n_perm = perm_lds ? PB + 64 * fa.w.n_cg : 0, n_woff = 2 * fa.w.n_cg + 1;
    staged = PLAIN || (a.n_pad <= TR * NT && n_perm <= PC * NT && n_woff <= NT && a.cs_nbk <= 64 && 2 * K <= NT / 64);   // (uniform; fused_prologue_staged on the host)
    if (staged) {
      float4 tlo[TR], thi[TR];
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = threadIdx.x + i * NT;
        const float* src = a.gw_s + (size_t)min(r, a.n_pad - 1) * KP;
        tlo[i] = *reinterpret_cast<const float4*>(src);
        thi[i] = K > 4 ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      int pv[PC];
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        const int j = threadIdx.x + i * NT;
        pv[i] = 0;
        if (j < n_perm) pv[i] = j < PB ? a.ell_pix[(size_t)blockIdx.x * PB + j] : fa.w.chan_perm[(size_t)blockIdx.x * fa.w.n_cg * 64 + (j - PB)];
      }
      int mh = 0, mw = 0;
      if ((int)threadIdx.x < 3 * NGRP) {
        const int gi = threadIdx.x / 3, j = threadIdx.x - 3 * gi;
        const bool tile_ok = blockIdx.x * PB + (gi >> GPT_SHIFT) * TP < a.p_pad;   // (an odd number of tiles: the last block has one)
        mh = tile_ok ? a.ell_off[2 * (blockIdx.x * NGRP + gi) + j] : 0;
      }
      if ((int)threadIdx.x < n_woff) mw = fa.w.ell_off[(size_t)2 * blockIdx.x * fa.w.n_cg + threadIdx.x];
      double csv = 0.0;
      const int cw = threadIdx.x >> 6;   // wave w: column sum w of G W' from the W update's partials; wave K + w (shared tail): sum of W'[:, w]
      if (cs_lds && cw < K && (int)(threadIdx.x & 63) < a.cs_nbk) csv = a.cs_parts[(size_t)cw * a.cs_nbk + (threadIdx.x & 63)];
      if (spread && cw >= K && cw < 2 * K && (int)(threadIdx.x & 63) < a.cs_nbk)
        csv = a.cs_parts[(size_t)K * a.cs_nbk + (size_t)(cw - K) * a.cs_nbk + (threadIdx.x & 63)];
      if (spread && rw_staged) {
#pragma unroll
        for (int u = 0; u < RW; ++u) {
          const int i = rw_lo + (int)threadIdx.x + u * NT;
          rwn[u] = i < rw_hi ? a.tail.w_new[i] : 1.f;
          rwo[u] = i < rw_hi ? a.tail.w_old[i] : 1.f;
        }
      }
      // ---- consume
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = threadIdx.x + i * NT;
        if (r < a.n_pad) FixTab<K>::put(tab, a.n_pad, r, tlo[i], thi[i]);
      }
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        const int j = threadIdx.x + i * NT;
        if (j < n_perm) lpix[j] = pv[i];   // (lchan = lpix + PB: one array)
      }
      if ((int)threadIdx.x < 3 * NGRP) meta[threadIdx.x] = mh;
      if ((int)threadIdx.x < n_woff) meta[3 * NGRP + threadIdx.x] = mw;
      if (cs_lds && cw < (spread ? 2 * K : K)) {
        csv = wave_sum(csv);
        if ((threadIdx.x & 63) == 0) (cw < K ? cs_lds : cs_lds + KP - K)[cw] = csv;   // (sums of W' behind the KP column sums)
      }
      if (threadIdx.x == 0) {
        cnt[0] = OWN_H ? NT / 64 : 0;    // (OWN_H: unit w of the H walk is wave w's)
        cnt[1] = PREF_W ? NT / 64 : 0;   // (PREF_W: channel group w is wave w's)
      }
    }
  }
  if (!PLAIN && !staged) {
    if (cs_lds) {      // wave w: column sums w, w + NT / 64, ... of G W' from the W update's partials (shared tail: then the sums of W' as well)
      for (int kk = threadIdx.x >> 6; kk < (spread ? 2 * K : K); kk += NT / 64) {
        double v = 0.0;
        for (int j = threadIdx.x & 63; j < a.cs_nbk; j += 64) v += a.cs_parts[(size_t)kk * a.cs_nbk + j];   // (row K + w of the partials: W'[:, w])
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) (kk < K ? cs_lds : cs_lds + KP - K)[kk] = v;
      }
    }
    for (int r = threadIdx.x; r < a.n_pad; r += NT) FixTab<K>::put_row(tab, a.n_pad, r, a.gw_s + (size_t)r * KP);
    if (threadIdx.x == 0) cnt[0] = cnt[1] = 0;
    // the block's list offsets, once: a unit then starts from LDS instead of from two dependent scalar loads
    if ((int)threadIdx.x < 3 * NGRP) {
      const int gi = threadIdx.x / 3, j = threadIdx.x - 3 * gi;
      const bool tile_ok = blockIdx.x * PB + (gi >> GPT_SHIFT) * TP < a.p_pad;   // (an odd number of tiles: the last block has one)
      meta[threadIdx.x] = tile_ok ? a.ell_off[2 * (blockIdx.x * NGRP + gi) + j] : 0;
    }
    for (int i = threadIdx.x; i <= 2 * fa.w.n_cg; i += NT) meta[3 * NGRP + i] = fa.w.ell_off[(size_t)2 * blockIdx.x * fa.w.n_cg + i];
    // What a unit needs before its first list row - the slot -> pixel map, the block's channel order - is fetched once by the
    // workgroup (below the full geometry always; at the full geometry where the 4 (PB + 64 n_cg) bytes fit, ESPM_FUSED_FULL_PERM_LDS),
    // so that a unit starts with ONE round trip to memory (its H column or GW rows and its first list rows together) instead
    // of a chain of three.
    if (perm_lds) {
      for (int i = threadIdx.x; i < PB; i += NT) lpix[i] = a.ell_pix[(size_t)blockIdx.x * PB + i];
      for (int i = threadIdx.x; i < 64 * fa.w.n_cg; i += NT) lchan[i] = fa.w.chan_perm[(size_t)blockIdx.x * fa.w.n_cg * 64 + i];
    }
  }
  __syncthreads();
  ESPM_PHASE_STAMP(1);
  float relw = -1.f;
  if (spread) {
    if (blockIdx.x == 0 && threadIdx.x < KP) a.tail.colsum_gw[threadIdx.x] = (int)threadIdx.x < K ? cs_lds[threadIdx.x] : 0.0;
    if (a.tail.hist_slot) {
      double sw = 0.0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) sw += cs_lds[KP + kk];
      const float shift = (float)((double)a.tail.rel_tol * (sw / (double)rw_n));   // tol * mean(W'), base.py:323
      relw = 0.f;
      if (rw_staged && (PLAIN || staged)) {
#pragma unroll
        for (int u = 0; u < RW; ++u)
          if (rw_lo + (int)threadIdx.x + u * NT < rw_hi) relw = fmaxf(relw, fabsf(rwn[u] - rwo[u]) / (rwn[u] + shift));
      } else {
        for (int i = rw_lo + (int)threadIdx.x; i < rw_hi; i += NT) {
          const float x = a.tail.w_new[i], y = a.tail.w_old[i];
          relw = fmaxf(relw, fabsf(x - y) / (x + shift));
        }
      }
    }
  }
  const int lane = threadIdx.x & 63;
  const int blk0 = blockIdx.x * PB;
  int own_next[2] = {(int)(threadIdx.x >> 6), (int)(threadIdx.x >> 6)};
  auto next_unit = [&](int which) {
    if (!PLAIN && fa.static_units) {
      const int u = own_next[which];
      own_next[which] += NT / 64;
      return __builtin_amdgcn_readfirstlane(u);
    }
    int u = 0;
    if (lane == 0) u = __hip_atomic_fetch_add(cnt + which, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(u);
  };

  // ---- H walk: units u = segment * 16 + group (segment-major: every group is started early) ----
  for (int u = OWN_H ? wave_id : next_unit(0); u < NGRP * S; u = next_unit(0)) {
    const int seg = u / NGRP, gi = u - seg * NGRP;     // group gi of the block: tile gi >> GPT_SHIFT
    const int grp = blk0 / 64 + gi;
    float acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[kk] = 0.f;
    float kl = 0.f;
    int lp = lane;   // place of the lane's pixel inside its tile
    if (blk0 + (gi >> GPT_SHIFT) * TP < a.p_pad) {   // (an odd number of tiles: the last block has one; the other's pixels lie beyond p)
      lp = perm_lds ? lpix[gi * 64 + lane] : a.ell_pix[grp * 64 + lane];   // slot -> pixel of the window (lists ordered by length)
      const int beg = meta[3 * gi], mid = meta[3 * gi + 1] - beg, len = meta[3 * gi + 2] - beg;
      const int x0 = FULL ? (S_RT ? FusedGeom<K>::seg_begin_rt(len, seg, S) : FusedGeom<K>::seg_begin(len, seg)) : FusedGeom<K>::seg_begin_even(len, seg, S);
      const int x1 = FULL ? (S_RT ? FusedGeom<K>::seg_begin_rt(len, seg + 1, S) : FusedGeom<K>::seg_begin(len, seg + 1)) : FusedGeom<K>::seg_begin_even(len, seg + 1, S);
      if (x0 < x1) {
        const int px = blk0 + (gi >> GPT_SHIFT) * TP + lp;
        float hk[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) hk[kk] = a.h_in[(size_t)kk * a.p_pad + px];
        const uint32_t* lrow = a.ell + (size_t)beg * 64 + lane;
        ell_h_rows<K, LOSS, UNR_H, PF, PRIO, STREAM, true, WALK_PP>(lrow, x0, x1, mid, tab, a.n_pad, a.ell_bits, hk, acc, kl);
        if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
      }
    }
    float* dst = part + (size_t)seg * PROWS * PB + (gi >> GPT_SHIFT) * TP + lp;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) dst[(size_t)kk * PB] = acc[kk];
    if constexpr (LOSS) {
      const float klw = wave_sum(kl);   // (a fixed butterfly: the same bits whoever walks the unit)
      if (lane == 0) ukl[u] = klw;
    } else if (lane == 0) {
      ukl[u] = 0.f;
    }
  }
  ESPM_PHASE_STAMP(2);   // wave 0 found no unit left
  ESPM_WAVE_STAMP(8);
  // per-pixel epilogue over the pixels of the block; H' rows go into the LDS table of the W walk (rows of the
  // pixels beyond p: ones, never referenced by an entry with a count)
  // the wave's first W unit (channel group = wave number), requested behind the pixels' update, ahead of the record reduction and its
  // barrier: the group's channels and list offsets come from LDS, its rows of G W and first list rows from memory
  uint32_t fw_rows[PF][UNR_W];
  float fw_gw[K];
  int fw_c = -1, fw_kind = 0;
  auto request_w = [&]() {
    if constexpr (PREF_W) {
      if (wave_id < fa.w.n_cg) {
        const int cg = wave_id;
        fw_c = perm_lds ? lchan[cg * 64 + lane] : fa.w.chan_perm[((size_t)blockIdx.x * fa.w.n_cg + cg) * 64 + lane];
        const float* gsrc = fa.w.gw_s + (size_t)(fw_c < 0 ? 0 : fw_c) * KP;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) fw_gw[kk] = gsrc[kk];
        const int* off = meta + 3 * NGRP + 2 * cg;
        const int beg = off[0], mid = off[1], end = off[2];
        const uint32_t* lrow = fa.w.ell + (size_t)beg * 64 + lane;
        if (ell_walk_request<UNR_W, PF, STREAM>(lrow, mid - beg, fw_rows)) fw_kind = 1;
        else if (mid == beg && ell_walk_request<UNR_W, PF, STREAM>(lrow, end - mid, fw_rows)) fw_kind = 2;
      }
    }
  };
  h_epilogue<K, true, 0, ESPM_FUSED_SUM_BATCHED ? (FULL ? FusedGeom<K>::S_MAX : ESPM_FUSED_MAX_SEGS) : 0, PLAIN>(
      a, part, S, PB, blk0, 0.f, cs_lds, tab, PB, false,
      (PLAIN || fa.red_lds_off >= 0) ? reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(smem) + fa.red_lds_off) : nullptr, relw, request_w,
      ukl, NGRP * S);

  ESPM_PHASE_STAMP(5);   // epilogue done (3: every wave has walked, 4: per-pixel work of wave 0 done - stamped inside h_epilogue)
  // ---- W accumulation: the block's channel groups, longest first (w_accum_ell_kernel's walk) ----
  const WAccumArgs& w = fa.w;
  const int b = blockIdx.x;
  // Units of the W walk: the block's channel groups, or - w_split, where the LDS holds TWO copies of the slab - their halves: the rows
  // of a group cut in two (unit rows and general rows each), half h adding up in copy h, the two copies summed at the write-out in that
  // order whoever walked them.  Why: 32 units on 16 waves leave nothing to hand to the wave that is done early, and the waves of a
  // SIMD are served oldest first - the W walk's youngest waves ended 5 us after wave 0 (profiles/r04c_phase_clock_512rows.log).
  const bool w_split = !PLAIN && ESPM_FUSED_W_SPLIT && fa.w_split;   // (uniform)
  const bool slab_lds = PLAIN || (ESPM_FUSED_SLAB_LDS && fa.slab_lds);   // (uniform)
  const int w_units = w_split ? 2 * w.n_cg : w.n_cg;
  bool first_w = PREF_W;   // (uniform) the unit in hand is the wave's own, requested ahead
  for (int wu = PREF_W ? wave_id : next_unit(1); wu < w_units; wu = next_unit(1), first_w = false) {
    const int cg = w_split ? wu >> 1 : wu, whalf = w_split ? wu & 1 : 0;
    const int c = (PREF_W && first_w) ? fw_c : (perm_lds ? lchan[cg * 64 + lane] : w.chan_perm[((size_t)b * w.n_cg + cg) * 64 + lane]);
    const float* gsrc = w.gw_s + (size_t)(c < 0 ? 0 : c) * KP;
    float gw[K], acc[K];
    if (PREF_W && first_w) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) gw[kk] = fw_gw[kk];
    } else {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) gw[kk] = gsrc[kk];
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[kk] = 0.f;
    const int wkind = (PREF_W && first_w) ? fw_kind : 0;
    const int* off = meta + 3 * NGRP + 2 * cg;
    const int beg = off[0], mid = off[1], end = off[2];
    int u0 = beg, u1 = mid, g0 = mid, g1 = end;   // unit rows [u0, u1), general rows [g0, g1) of this unit
    if (w_split) {
      const int hu = beg + (mid - beg) / 2, hg = mid + (end - mid) / 2;
      if (whalf == 0) {
        u1 = hu;
        g1 = hg;
      } else {
        u0 = hu;
        g0 = hg;
      }
    }
    const uint32_t* lrow = w.ell + (size_t)beg * 64 + lane;
    ell_walk_pre<K, UNR_W, PF, PRIO, STREAM, WALK_PP>(lrow + (size_t)(u0 - beg) * 64, u1 - u0, EllGetUnitFix<K>(PB), [&](float, const float (&h)[K]) {
      ell_axpy<K>(acc, h, __builtin_amdgcn_rcpf(ell_dot<K>(h, gw)));
    }, EllNoFlush(), wkind == 1, fw_rows);
    ell_walk_pre<K, UNR_W, PF, PRIO, STREAM, WALK_PP>(lrow + (size_t)(g0 - beg) * 64, g1 - g0, EllGetFix<K>(tab, PB, PBITS), [&](float x, const float (&h)[K]) {
      const float r = x * __builtin_amdgcn_rcpf(ell_dot<K>(h, gw));
      ell_axpy<K>(acc, h, r);
    }, EllNoFlush(), wkind == 2, fw_rows);
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
    if (c >= 0) {
      if (slab_lds) {
        float* copy = part + (size_t)whalf * K * w.n_pad;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) copy[(size_t)kk * w.n_pad + c] = acc[kk];
      } else {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) w.a_slab[((size_t)b * K + kk) * w.n_pad + c] = acc[kk];
      }
    }
  }
  ESPM_PHASE_STAMP(6);   // wave 0 found no channel group left
  ESPM_WAVE_STAMP(24);
  if (slab_lds) {
    __syncthreads();
    // rows of the slab, 16 bytes per store, write-through (sc0 sc1: the line leaves the XCD's L2 now, not at the end of the launch);
    // the entries of the channels n .. n_pad - 1 belong to no list: zeros, as the scattered stores left them
    float* dst = w.a_slab + (size_t)b * K * w.n_pad;
    typedef float xf4 __attribute__((ext_vector_type(4)));
    for (int i = threadIdx.x; i < (K * w.n_pad) / 4; i += NT) {
      xf4 v = reinterpret_cast<const xf4*>(part)[i];
      if (w_split) v += reinterpret_cast<const xf4*>(part + (size_t)K * w.n_pad)[i];   // (copy 0 + copy 1, in this order)
      const int c0 = (4 * i) % w.n_pad;
      if (c0 + 3 >= a.n) {
        if (c0 >= a.n) v[0] = 0.f;
        if (c0 + 1 >= a.n) v[1] = 0.f;
        if (c0 + 2 >= a.n) v[2] = 0.f;
        v[3] = 0.f;
      }
      // (s_nop: the hardware wants wait states between a store of more than 64 bits and a write of its data registers, which
      //  the compiler cannot place for an instruction it does not see)
#if ESPM_FUSED_SLAB_WT
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst + 4 * i), "v"(v) : "memory");
#else
      *reinterpret_cast<xf4*>(dst + 4 * i) = v;
#endif
    }
  }
#ifdef ESPM_PHASE_CLOCK
  __syncthreads();
  ESPM_PHASE_STAMP(7);   // the workgroup is done
#endif
}

}  // namespace espm
