// Launcher of the fused half-steps of the sparse count store (mu_fused_kernel.hpp).
#include <stdlib.h>

#include "mu_fused_kernel.hpp"

#ifndef ESPM_ELL_UNR_H
#define ESPM_ELL_UNR_H 4
#endif
#ifndef ESPM_FUSED_SMALL_UNR_H   // list dwords per batch in the run-time-sized instance (shards, small images): A/B knob (code size against rows in flight)
#define ESPM_FUSED_SMALL_UNR_H ESPM_ELL_UNR_H
#endif
#ifndef ESPM_FUSED_SMALL_UNR_W
#define ESPM_FUSED_SMALL_UNR_W ESPM_ELL_UNR_W
#endif
#ifndef ESPM_ELL_UNR_W
#define ESPM_ELL_UNR_W 4
#endif

// the full geometry: the block's permutations in LDS (mu_fused_kernel.hpp) when they fit next to the tables.  Measured: no
// gain at the headline (150.1 against 146.9-150.2 us per iteration, profiles/r03a_variant_ab_512.log: with four waves per SIMD
// a unit's start - 4 of them per wave and walk - is covered by the other waves) - off.
#ifndef ESPM_FUSED_FULL_PERM_LDS
#define ESPM_FUSED_FULL_PERM_LDS 0
#endif
#define ESPM_FUSED_LDS_LIMIT (160 * 1024)   // a workgroup's LDS on gfx950
#ifndef ESPM_FUSED_SMALL_SEGS_DEFAULT       // below the full geometry: segments per list group where the LDS holds them (0: 1024 / pb)
#define ESPM_FUSED_SMALL_SEGS_DEFAULT 0
#endif

#ifndef ESPM_FUSED_PLAIN   // the lean instance of the common case (mu_fused_plain.hip); ESPM_FUSED_PLAIN=0 in the environment keeps the generic one (A/B)
#define ESPM_FUSED_PLAIN 1
#endif

namespace espm {

// The A/B switches of the environment, read ONCE per process (a lookup per launch sat inside espm_mu_iterate's loop, raced
// with a setenv from another thread and could change the kernel instance in the middle of a fit).
static bool fused_env_plain() {
  static const bool on = [] {
    const char* env = getenv("ESPM_FUSED_PLAIN");
    return !(env && env[0] == '0');
  }();
  return on;
}
static int fused_env_small_segs() {
  static const int want = [] {
    const char* env = getenv("ESPM_FUSED_SMALL_SEGS");
    return env ? atoi(env) : ESPM_FUSED_SMALL_SEGS_DEFAULT;
  }();
  return want;
}

int launch_fused_plain(const FusedArgs& args, int k, bool loss, bool full, int nblk, size_t lds_bytes, hipStream_t stream);   // mu_fused_plain.hip
int launch_fused_plain_stream(const FusedArgs& args, int k, bool loss, bool full, int nblk, size_t lds_bytes, hipStream_t stream);   // mu_fused_stream.hip
#ifdef ESPM_PHASE_CLOCK
int phase_buffer_plain(unsigned long long* p);
int phase_buffer_plain_stream(unsigned long long* p);
#endif

#if ESPM_MIN_K <= 8
template <int K>
static int launch_fused_k(const FusedArgs& args_in, int nblk, hipStream_t stream) {
  FusedArgs args = args_in;
  const int pb = args.w.pb;
  const int tab_rows = args.h.n_pad > pb ? args.h.n_pad : pb;
  const bool full = pb == ESPM_ELL_PB;
  ESPM_REQUIRE(K <= 4 || tab_rows <= FixTab<K>::MAX_ROWS, "fused half-steps: a table of %d rows, at most %d from 5 components on", tab_rows, FixTab<K>::MAX_ROWS);
  const size_t red = (size_t)(ESPM_ELL_WTHREADS / 64 + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  const size_t tail_scratch = (size_t)((ESPM_ELL_WTHREADS / 64 + 1) * (KP + 1) + 1) * sizeof(double);
  const size_t red_own = (size_t)(ESPM_ELL_WTHREADS / 64) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);   // the one-barrier record reduction's scratch
  auto part_of = [&](int segs) {
    size_t part = (size_t)segs * FusedGeom<K>::PROWS * pb * sizeof(float);
    if (red > part) part = red;
    if (tail_scratch > part) part = tail_scratch;
    return part;
  };
  size_t bytes = 0;
  // the workgroup's LDS with `part` bytes for the numerators' region.  red_in_table: the scratch of the record reduction in the part of the
  // table region that is dead by then - the G W table is read by the H walk only, the H' table of the W walk occupies the first pb rows'
  // worth of it (mu_fused_kernel.hpp) - where the region is long enough; otherwise (and by default) behind the permutations
  auto layout = [&](size_t part, bool red_in_table) {
    const size_t tab_bytes = FixTab<K>::bytes(tab_rows);   // (float4 parts from address 0, components 4.. from ESPM_TAB2_BASE: mu_h_kernel.hpp)
    bytes = tab_bytes + part;
    args.cnt_lds_off = (int)bytes;   // the two unit counters, then the units' KL sums
    bytes += 16 + ESPM_FUSED_MAX_UNITS * sizeof(float);
    args.meta_lds_off = (int)bytes;  // the block's list offsets
    bytes += (size_t)(3 * (pb / 64) + 2 * args.w.n_cg + 1 + 3) / 4 * 16;
    args.perm_lds_off = (int)bytes;  // below the full geometry: the block's pix_perm and chan_perm
    const size_t perm_bytes = (size_t)(pb + 64 * args.w.n_cg) * sizeof(int);
    // (the full geometry: the copies are an extra where the workgroup's 160 KB still hold them)
    args.perm_lds = pb != ESPM_ELL_PB || (ESPM_FUSED_FULL_PERM_LDS && bytes + perm_bytes + KP * sizeof(double) <= ESPM_FUSED_LDS_LIMIT);
    if (args.perm_lds) bytes += perm_bytes;
    args.red_lds_off = -1;
    const size_t hp_tab = FixTab<K>::bytes(pb);   // (the end of the H' table's last part: what lies behind it up to tab_bytes held rows pb.. of G W)
    if (ESPM_FUSED_RED_ONE_BARRIER && red_in_table && hp_tab + red_own <= tab_bytes) {
      args.red_lds_off = (int)hp_tab;
    } else if (ESPM_FUSED_RED_ONE_BARRIER && bytes + red_own + 2 * KP * sizeof(double) <= ESPM_FUSED_LDS_LIMIT) {
      bytes = (bytes + 7) / 8 * 8;
      args.red_lds_off = (int)bytes;
      bytes += red_own;
    }
    if (args.h.cs_parts) {   // the workgroup's own copy of colsum(GW'): k doubles behind the numerators (and, for the shared tail, the sums of W')
      bytes = (bytes + 7) / 8 * 8;
      args.h.cs_lds_off = (int)bytes;
      bytes += 2 * KP * sizeof(double);
    }
  };
  // segments per list group of the H walk: below the full geometry 1024 / pb (16 units); at the full geometry as many as fit, S_MAX down to S
  args.h_segs = full ? FusedGeom<K>::S : ESPM_ELL_PB / pb;
  size_t part0 = part_of(args.h_segs);
  bool red_in_table = false;
  if (full) {
    for (int segs = FusedGeom<K>::S_MAX; segs > FusedGeom<K>::S; --segs) {
      bool ok = false;
      for (int rit = 0; rit < 2 && !ok; ++rit) {
        layout(part_of(segs), rit != 0);
        ok = bytes <= ESPM_FUSED_LDS_LIMIT && args.red_lds_off >= 0;
        if (ok) red_in_table = rit != 0;
      }
      if (ok) {
        args.h_segs = segs;
        part0 = part_of(segs);
        break;
      }
    }
  }
  // the slab of the block collected in the numerators' region (mu_fused_kernel.hpp, ESPM_FUSED_SLAB_LDS): needs that region to itself from
  // the epilogue's barrier on (the record reduction's scratch elsewhere) and k n_pad floats of it - below the full geometry the
  // region is grown to that where the workgroup's LDS allows
  const size_t slab = (size_t)K * args.w.n_pad * sizeof(float);
  args.slab_lds = 0;
  layout(part0, red_in_table);
  args.w_split = 0;
  if (ESPM_FUSED_SLAB_LDS && args.w.n_pad % 4 == 0) {
    if (slab > part0) {
      layout(slab, false);
      if (bytes > ESPM_FUSED_LDS_LIMIT || args.red_lds_off < 0) {
        layout(part0, red_in_table);
      } else {
        args.slab_lds = 1;
        // the region grown for the slab holds more partial sets than 1024 / pb: more (group, segment) units than waves, so that the
        // units handed out last level the waves (ESPM_FUSED_SMALL_SEGS: A/B; 0 keeps 1024 / pb)
        const int want = fused_env_small_segs();
        const int fit = (int)(slab / ((size_t)FusedGeom<K>::PROWS * pb * sizeof(float)));
        int segs = want > 0 ? want : args.h_segs;
        if (segs > fit) segs = fit;
        if (segs > ESPM_FUSED_MAX_SEGS) segs = ESPM_FUSED_MAX_SEGS;
        if (segs > args.h_segs) args.h_segs = segs;
      }
    } else {
      args.slab_lds = args.red_lds_off >= 0;
      // the full geometry: two copies of the slab where the region holds them - the W walk then hands out half channel groups
      args.w_split = ESPM_FUSED_W_SPLIT && args.slab_lds && pb == ESPM_ELL_PB && 2 * slab <= part0;
    }
  }
  // the tail of the previous W update is shared by the launch's own workgroups (mu_fused_kernel.hpp) unless it carries the
  // projected gradient's quadratic term, which only the extra workgroup sums
  if (args.h.tail_on && !args.h.tail.pg_q) args.h.tail_on = 2;
  if (bytes > ((args.perm_lds && pb == ESPM_ELL_PB) || args.red_lds_off >= 0 ? ESPM_FUSED_LDS_LIMIT : ESPM_ELL_LDS_MAX))
    return set_error(ESPM_EUNSUPPORTED, "fused half-steps: %zu bytes of LDS exceed %d", bytes, ESPM_ELL_LDS_MAX);
  // the common case as its own, leaner instance (mu_fused_kernel.hpp: PLAIN; instantiated in mu_fused_plain.hip): every fact the
  // kernel takes for granted is checked here
  {
    const int NT = ESPM_ELL_WTHREADS;
    const int n_perm = args.perm_lds ? pb + 64 * args.w.n_cg : 0, n_woff = 2 * args.w.n_cg + 1;
    const bool staged_ok = args.h.n_pad <= 4 * NT && n_perm <= 4 * NT && n_woff <= NT && args.h.cs_nbk <= 64 && 2 * K <= NT / 64;
    const bool plain = ESPM_FUSED_PLAIN && fused_env_plain() && !args.h.fixed_h && !args.h.fill_num && !args.h.breg_sr && !args.h.l2_m &&
                       args.h.simplex_h && args.h.lambda_l != 0.f && args.h.grid_mode && args.h.have_prev && args.h.write_h && args.h.h_rule == 0 &&
                       args.h.tail_on != 1 && !args.static_units && args.slab_lds && !args.w_split && args.red_lds_off >= 0 && staged_ok &&
                       (args.perm_lds != 0) == (pb != ESPM_ELL_PB) && ESPM_FUSED_SMALL_THREADS == ESPM_ELL_WTHREADS;
    // (the lists too large for the last-level cache, the caller says: the instance that loads them without allocating there)
    if (plain && args.stream_lists && pb == ESPM_ELL_PB) return launch_fused_plain_stream(args, K, args.h.compute_loss != 0, true, nblk, bytes, stream);
    if (plain) return launch_fused_plain(args, K, args.h.compute_loss != 0, pb == ESPM_ELL_PB, nblk, bytes, stream);
  }
  auto go = [&](auto kern, int threads) -> int {
    if (bytes > 64 * 1024)
      if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                             "fused half-steps"))
        return rc;
    hipLaunchKernelGGL(kern, dim3(nblk + (args.h.tail_on == 1 ? 1 : 0)), dim3(threads), bytes, stream, args);
    return check_hip(hipGetLastError(), "fused half-steps launch");
  };
  if (pb == ESPM_ELL_PB)   // the full geometry: 16 waves, sizes known at compile time
    return args.h.compute_loss ? go(mu_fused_ell_kernel<K, true, ESPM_ELL_UNR_H, ESPM_ELL_UNR_W, ESPM_ELL_WTHREADS, true>, ESPM_ELL_WTHREADS)
                               : go(mu_fused_ell_kernel<K, false, ESPM_ELL_UNR_H, ESPM_ELL_UNR_W, ESPM_ELL_WTHREADS, true>, ESPM_ELL_WTHREADS);
  return args.h.compute_loss ? go(mu_fused_ell_kernel<K, true, ESPM_FUSED_SMALL_UNR_H, ESPM_FUSED_SMALL_UNR_W, ESPM_FUSED_SMALL_THREADS, false>, ESPM_FUSED_SMALL_THREADS)
                             : go(mu_fused_ell_kernel<K, false, ESPM_FUSED_SMALL_UNR_H, ESPM_FUSED_SMALL_UNR_W, ESPM_FUSED_SMALL_THREADS, false>, ESPM_FUSED_SMALL_THREADS);
}
#endif

// LDS the fused kernel needs for (n_pad, k): the caller decides with it whether the fused path applies
size_t fused_ell_lds_bytes(int n_pad, int k, int pb) {
  if (pb < 128 || pb > ESPM_ELL_PB || (pb & (pb - 1))) return (size_t)-1;
  const int tab_rows = n_pad > pb ? n_pad : pb;
  const bool fixed = k > 4 && k >= ESPM_FIXTAB_MIN_K;   // (FixTab, mu_h_kernel.hpp)
  if (fixed && tab_rows > ESPM_TAB2_BASE / 16) return (size_t)-1;   // (those tables hold at most 2048 rows)
  const int wb = k <= 4 ? 0 : (k == 5 ? 1 : (k == 6 ? 2 : 4));
  const size_t tab_bytes = fixed ? (size_t)ESPM_TAB2_BASE + (size_t)16 * tab_rows : (size_t)(4 + wb) * 4 * tab_rows;
  const int seg = pb == ESPM_ELL_PB ? (k <= 4 ? 4 : (k <= 6 ? 3 : 2)) : ESPM_ELL_PB / pb;   // FusedGeom<K>::S | 1024 / pb
  const int n_cg = (n_pad + 63) / 64;   // (>= the channel groups of any n with this n_pad)
  size_t part = (size_t)seg * k * pb * 4;   // (FusedGeom<K>::PROWS = K: the units' KL sums are ESPM_FUSED_MAX_UNITS floats of their own)
  const size_t red = (size_t)(ESPM_ELL_WTHREADS / 64 + 1) * (ESPM_HP_NSCALAR + 2 * k + 1) * sizeof(double);
  const size_t tail_scratch = (size_t)((ESPM_ELL_WTHREADS / 64 + 1) * (KP + 1) + 1) * sizeof(double);
  if (red > part) part = red;
  if (tail_scratch > part) part = tail_scratch;
  return tab_bytes + part + 16 + ESPM_FUSED_MAX_UNITS * sizeof(float) + (size_t)(3 * (pb / 64) + 2 * n_cg + 4) / 4 * 16 + 2 * KP * sizeof(double) + 8 +
         (pb != ESPM_ELL_PB ? (size_t)(pb + 64 * n_cg) * sizeof(int) : 0);
}

int launch_fused_ell(const HStepArgs& h, const WAccumArgs& w, int nblk, hipStream_t stream, int static_units, int stream_lists) {
  ESPM_REQUIRE(h.ell && h.ell_off && h.ell_klc && h.ell_pix && w.ell && w.ell_off && w.chan_perm, "fused half-steps: the sparse store's lists are missing");
  ESPM_REQUIRE(w.pb >= 128 && w.pb <= ESPM_ELL_PB && (w.pb & (w.pb - 1)) == 0 && 2 * h.ell_tp == w.pb && h.h_rule == 0 && h.write_h && !h.l2_m,
               "fused half-steps: blocks of two H tiles (tile_px=%d, ell_pb=%d), the default H rule, write_h", h.ell_tp, w.pb);
  ESPM_REQUIRE(nblk == (h.p + w.pb - 1) / w.pb && w.n_cg >= 1, "fused half-steps: nblk_w=%d must be ceil(p / %d)", nblk, w.pb);
  FusedArgs fa;
  fa.h = h;
  fa.w = w;
  fa.static_units = static_units;
  fa.stream_lists = stream_lists;
  switch (h.k) {
#if ESPM_MIN_K <= 8
#define ESPM_X(KK) case KK: return launch_fused_k<KK>(fa, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
  }
  return set_error(ESPM_EUNSUPPORTED, "fused half-steps: k=%d not built", h.k);
}

}  // namespace espm

#ifdef ESPM_PHASE_CLOCK
// debug build only (tools/analysis/phase_clock.py): where the fused kernel's workgroups write their phase stamps (8 x uint64 per workgroup)
extern "C" int espm_debug_phase_buffer(void* dev_ptr) {
  unsigned long long* p = static_cast<unsigned long long*>(dev_ptr);
  if (int rc = espm::phase_buffer_plain(p)) return rc;
  if (int rc = espm::phase_buffer_plain_stream(p)) return rc;
  return espm::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(espm::espm_phase_buf), &p, sizeof(p)), "phase buffer");
}
#endif
