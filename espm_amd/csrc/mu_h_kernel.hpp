// The H-step kernel template (see mu_h_step.hip for the description); in a header so that the tuning
// harness (tools/tune) can instantiate variants next to the product's dispatch table.
#pragma once
// the wave stage of the fused kernel's record reduction through gfx950's half / row exchanges (mu_common.hpp, wave_reduce_packed)
// H' of the fused launch stored write-through (sc1), so that the launch does not end with its 4 k p bytes dirty in the L2s (0: plain stores, A/B)
#ifndef ESPM_HOUT_WRITE_THROUGH
#define ESPM_HOUT_WRITE_THROUGH 1
#endif
#ifndef ESPM_FUSED_RED_PACKED
#define ESPM_FUSED_RED_PACKED 1
#endif
#include "mu_common.hpp"

namespace espm {

// Row r of an LDS gather table of `rows` rows of K floats, in the layout of EllTab<K> (mu_ell_kernel.hpp): components 4..
// first (1, 2 or 4 floats per row), then the float4 part.
template <int K>
struct LdsTabGeom {
  // floats of a row beyond the float4 part: 0, 1, 2, 4 up to 8 components; 8 (two float4) up to 12, 12 (three) up to 16
  static constexpr int WB = K <= 4 ? 0 : (K == 5 ? 1 : (K == 6 ? 2 : (K <= 8 ? 4 : (K <= 12 ? 8 : 12))));
};
// row r from the first K of `src` (a KP-strided row of gw_s / h_t, or registers)
template <int K>
__device__ __forceinline__ void lds_table_put_row(float* tab, int rows, int r, const float* src) {
  constexpr int WB = LdsTabGeom<K>::WB;
  reinterpret_cast<float4*>(tab + (size_t)WB * rows)[r] = make_float4(src[0], K > 1 ? src[1] : 0.f, K > 2 ? src[2] : 0.f, K > 3 ? src[3] : 0.f);
  if constexpr (WB == 1) tab[r] = src[4];
  if constexpr (WB == 2) reinterpret_cast<float2*>(tab)[r] = make_float2(src[4], src[5]);
  if constexpr (WB >= 4) {
#pragma unroll
    for (int q = 0; q < WB / 4; ++q)
      reinterpret_cast<float4*>(tab)[(size_t)r * (WB / 4) + q] =
          make_float4(4 + 4 * q < K ? src[4 + 4 * q] : 0.f, 5 + 4 * q < K ? src[5 + 4 * q] : 0.f, 6 + 4 * q < K ? src[6 + 4 * q] : 0.f, 7 + 4 * q < K ? src[7 + 4 * q] : 0.f);
  }
}
template <int K>
__device__ __forceinline__ void lds_table_put(float* tab, int rows, int r, const float4 lo, const float4 hi) {
  static_assert(K >= 1, "");
  if constexpr (K <= 8) {
    const float src[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    lds_table_put_row<K>(tab, rows, r, src);
  }   // (more than 8 components: the fused kernel, the only caller with two float4, is not built for them)
}

// ---- the FUSED kernel's gather tables (round 5): both parts of a row behind ONE address register ------------------------------------------
// The layout above puts components 4.. first (1, 2 or 4 floats per row) so that a unit entry - the byte offset 16 r of its row's float4 -
// reaches them with a shift; the float4 part then sits behind them, at a base that depends on the rows of the table (a run-time value in
// the H walk, and in both walks below the full geometry): one vector add per entry, one shift or bit-field extract for the other address -
// two of a unit entry's 11 (k = 5) .. 14 (k = 8) vector instructions.  Here the float4 part of row r is at byte 16 r from LDS address 0
// and components 4.. at byte ESPM_TAB2_BASE + 16 r - a COMPILE-TIME distance, so both reads take the entry's 16 bits as their address
// register and differ in the instruction's offset field: ONE vector instruction (and / shift) decodes a unit entry.  Rows of the second
// part are 16 bytes whatever k (2 or 4 floats used): 32 bytes of LDS per row (k = 6: 64 KB instead of 48 for 2048
// rows - the room is there since the KL part left the partials), at most ESPM_TAB2_BASE / 16 = 2048 rows (the launcher checks; larger
// tables keep the two-launch kernels and their layout).
#define ESPM_TAB2_BASE 32768
// MEASURED (one box, tools/analysis/variant_ab.py, profiles/r05j_ab_*.log; layout above | this one, us per iteration): k = 6 headline image
// 141.0-141.8 | 138.3; k = 8 161.4 | 159.5-159.9; configuration 5's 128-row shard 94.8 | 92.5-92.9 (its whole image 643 -> 638).  But k = 5:
// 131.6-131.9 | 140.4 at the headline, 30.8 | 32.0 on a 64-row shard, 41.7 | 45.0 on 128 rows - the fifth component's 4-byte reads at a
// 16-byte stride meet in a quarter of the banks (two passes each, by construction of the unit rows' placement), the table grows from 40 to
// 64 KB, and at k = 5 the LDS array, not the saved vector instruction, decides.  (The fifth component through an 8- or 16-byte read instead - 2-way and
// conflict-free by the same placement - was tried too: 131.8 | 134.1 | 138.7 us, profiles/r05s_ab_*.log.)  Hence from 6 components on.
#ifndef ESPM_FIXTAB_MIN_K   // component counts from which the fused kernel's tables take this layout (A/B: 9 keeps the layout above for all)
#define ESPM_FIXTAB_MIN_K 6
#endif
template <int K>
struct FixTab {
  static constexpr bool TWO = K > 4;
  static constexpr bool FIXED = K >= ESPM_FIXTAB_MIN_K;   // (K <= 4: one part either way)
  static constexpr int MAX_ROWS = (TWO && FIXED) ? ESPM_TAB2_BASE / 16 : (1 << 30);
  __host__ __device__ static constexpr size_t bytes(int rows) {
    return (TWO && FIXED) ? (size_t)ESPM_TAB2_BASE + 16u * (size_t)rows : (size_t)(4 + LdsTabGeom<K>::WB) * 4u * (size_t)rows;
  }
  static __device__ __forceinline__ void put(float* tab, int rows, int r, const float4 lo, const float4 hi) {
    if constexpr (!(TWO && FIXED)) {
      lds_table_put<K>(tab, rows, r, lo, hi);
    } else {
      reinterpret_cast<float4*>(tab)[r] = lo;
      float* p2 = tab + ESPM_TAB2_BASE / 4 + 4 * (size_t)r;
      if constexpr (K == 5) p2[0] = hi.x;
      else if constexpr (K == 6) *reinterpret_cast<float2*>(p2) = make_float2(hi.x, hi.y);
      else *reinterpret_cast<float4*>(p2) = hi;
    }
  }
  static __device__ __forceinline__ void put_row(float* tab, int rows, int r, const float* src) {   // src: a KP-strided row of gw_s
    const float4 lo = *reinterpret_cast<const float4*>(src);
    float4 hi = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (TWO && KP > 4) hi = *reinterpret_cast<const float4*>(src + 4);
    put(tab, rows, r, lo, hi);
  }
};

// What the epilogue needs of one pixel besides its numerators: requested in one go (no branch between the
// loads, so they are all in flight together) BEFORE the barrier that ends the accumulation phase - a wave that
// finishes its part early has them by the time the slowest wave arrives.
template <int K>
struct HEpiIn {
  float hin[K], hprev[K];
  float l[K], r[K], u[K], d[K];   // the four neighbours (raw: combined after the barrier, so that nothing waits for them before it)
  float wl, wr, wu, wd;           // 1 where the neighbour exists, else 0
};
template <int K, bool PLAIN = false>
__device__ __forceinline__ void h_epilogue_load(const HStepArgs& a, int q, bool stencil, HEpiIn<K>& v) {
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    v.hin[kk] = a.h_in[(size_t)kk * a.p_pad + q];
    v.hprev[kk] = (PLAIN || a.have_prev) ? a.h_out[(size_t)kk * a.p_pad + q] : 0.f;
    v.l[kk] = v.r[kk] = v.u[kk] = v.d[kk] = 0.f;
  }
  v.wl = v.wr = v.wu = v.wd = 0.f;
  if (stencil) {
    // 5-point Laplacian (utils.py:39-76): the four neighbours through clamped addresses and 0/1 weights; rows
    // above / below the local block come from the halo rows when present (sharded image)
    const int i = q / a.ny, j = q - i * a.ny;
    v.wl = j > 0 ? 1.f : 0.f;
    v.wr = j < a.ny - 1 ? 1.f : 0.f;
    const bool up_in = i > 0, dn_in = i < a.nx - 1;
    v.wu = (up_in || a.halo_top) ? 1.f : 0.f;
    v.wd = (dn_in || a.halo_bot) ? 1.f : 0.f;
    const int ql = j > 0 ? q - 1 : q, qr = j < a.ny - 1 ? q + 1 : q;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      const float* hrow = a.h_in + (size_t)kk * a.p_pad;
      const float* pu = up_in ? hrow + (q - a.ny) : (a.halo_top ? a.halo_top + (size_t)kk * a.ny + j : hrow + q);
      const float* pd = dn_in ? hrow + (q + a.ny) : (a.halo_bot ? a.halo_bot + (size_t)kk * a.ny + j : hrow + q);
      v.l[kk] = hrow[ql];
      v.r[kk] = hrow[qr];
      v.u[kk] = *pu;
      v.d[kk] = *pd;
    }
  }
}

// Epilogue shared by the H-step kernels: `smem` holds `nparts` partial numerators [part][K][TP] (written by
// the caller, not yet synchronised); one thread per pixel adds the regularisation terms, solves the simplex
// multiplier, clamps, writes H', H'^T and the per-workgroup record.  kl_lane = this lane's part of
// sum X log2(X / Y).
// EARLY = false requests the pixel's inputs after the barrier (for a kernel that cannot spare the registers across it).
// RULE: the H rule (a.h_rule) - 0 log surrogate, 1 quadratic surrogate, 2 projected gradient - as a compile-time switch,
// so that the default rule does not carry the registers of the others.
// lds_tab (fused half-steps, mu_fused_kernel.hpp): the row H'[:, pixel] goes into this LDS table of `lds_rows` rows (the W
// walk's gather table: row = place of the pixel in the tile, ones for the pixels beyond p) instead of the copy h_t in memory.
// kl_rows (fused half-steps): every partial carries K + 1 rows of TP floats, the last being the pixel's part of
// sum X log2(X / Y) (summed per pixel in slot order: the loss does not depend on which wave walked which slot); kl_lane
// is then unused and the per-pixel loss constant ell_klc is added here.
// MAXP > 0: at most MAXP partials - their LDS reads are then requested together; the sum keeps the slot order.  (Measured: no gain.  What
// a 64-row shard's update spends between its barrier and the sums - 2.1 us - is not the wait for these reads, nor for the pixel's inputs
// from memory (staging those in LDS in the prologue made it 3.1): the once-executed code of this kernel is fetched from L2 by every CU at
// every launch, and more unrolled code is more of that: profiles/r03c / r03d / r03e_phase_clock_64rows.log.  The callers pass 0.)
// relw_lane: this lane's share of rel_W of the W update that produced the input state (record field ESPM_HP_RELW), -1: none.
// PLAIN (the fused kernel's common case, mu_fused_plain.hip): what the launcher has checked on the host becomes a compile-time fact -
// simplex over H, Laplacian on an image grid, a previous H to compare with, H' written; no fixed_H, no fill numerators, neither
// the Bregman nor the Frobenius variant (mu stays a run-time flag: one uniform branch, and configuration 5 - mu = 0.05 - is a common case too).  The generic instance keeps ~100 scalar registers of flags and pointers alive through every
// phase (107-165 of them spilled to vector lanes) and walks their branches in 16 waves that share one scalar unit.
#ifndef ESPM_SIMPLEX_FAST_EXIT   // the per-pixel simplex root without its confirming evaluation (mu_common.hpp: simplex_root, fast_exit)
#define ESPM_SIMPLEX_FAST_EXIT 1
#endif
#ifndef ESPM_SUM_ROW_FENCE
#define ESPM_SUM_ROW_FENCE 1
#endif
#ifndef ESPM_SUM_SWITCH
#define ESPM_SUM_SWITCH 1
#endif
struct HEpiNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_pixels(): called by every thread behind its pixels' update, ahead of the record reduction and its barrier (the fused kernel
// requests the first rows of its W walk there).
template <int K, bool EARLY = true, int RULE = 0, int MAXP = 0, bool PLAIN = false, typename Hook = HEpiNoHook>
__device__ __forceinline__ void h_epilogue(const HStepArgs& a, float* smem, int nparts, int TP, int tile0, float kl_lane,
                                           const double* colsum = nullptr,   // the workgroup's own copy of colsum(GW) (LDS), else a.colsum_gw
                                           float* lds_tab = nullptr, int lds_rows = 0, bool kl_rows = false,
                                           double* red_scratch = nullptr,   // fused half-steps: scratch of its own for the waves' sums -> ONE barrier after the per-pixel work
                                           float relw_lane = -1.f, Hook after_pixels = Hook(),
                                           const float* unit_kl = nullptr, int n_unit_kl = 0) {   // fused half-steps (round 5): the H walk's units leave their KL sums as n_unit_kl floats in LDS instead of a row of the partials
  constexpr int NRED = ESPM_HP_NSCALAR + 2 * K + 1;  // sums: scalars (but RELH) + K row sums; max: RELH + K row maxima + RELW
  float red[NRED];   // per-thread partials in fp32 (one or two pixels per thread); fp64 from the wave results on (block_reduce_f32)
#pragma unroll
  for (int i = 0; i < NRED; ++i) red[i] = 0.f;
  // layout inside red[]: [0..3] KL, REG, LAP, BAD (sums), [4..4+K) row sums, [4+K] RELH, [5+K..5+2K) maxima
  constexpr int R_ROWSUM = 4, R_RELH = 4 + K, R_MAX = 5 + K, R_RELW = 5 + 2 * K;
  red[ESPM_HP_KL] = kl_lane;
  red[R_RELW] = relw_lane;
  double pg_q = 0.0;   // rule 2: <H' - H, grad> + gamma ||H' - H||^2 of this thread's pixels (the linesearch's quadratic bound)
  static_assert(!PLAIN || RULE == 0, "the plain instance: the default H rule");
  // the features as facts (PLAIN) or as the run-time flags they are
  const bool f_fill = PLAIN ? false : a.fill_num != nullptr, f_breg = PLAIN ? false : a.breg_sr != nullptr, f_l2 = PLAIN ? false : a.l2_m != nullptr;
  const bool f_mu = a.mu != nullptr, f_fixed = PLAIN ? false : a.fixed_h != nullptr;
  const bool f_prev = PLAIN ? true : a.have_prev != 0, f_lap = PLAIN ? true : a.lambda_l != 0.f, f_grid = PLAIN ? true : a.grid_mode != 0;
  const bool f_write = PLAIN ? true : a.write_h != 0, f_simplex = PLAIN ? true : a.simplex_h != 0;
  const bool stencil = f_lap && f_grid;
  HEpiIn<K> in;
  bool loaded = false;
  if (EARLY && K <= 6 && (int)threadIdx.x < TP && tile0 + (int)threadIdx.x < a.p) {   // (k = 7, 8: too many registers to hold across the barrier)
    h_epilogue_load<K, PLAIN>(a, tile0 + (int)threadIdx.x, stencil, in);
    loaded = true;
  }
  // likewise ahead of the barrier: what every pixel needs of the state's global statistics (scalar loads the compiler may
  // not move across a barrier itself) and the pixel's loss constant / fill mark
  const bool want_klc = kl_rows || f_fill || unit_kl != nullptr;
  float klc_first = 0.f;
  bool klc_loaded = false;
  if (want_klc && (int)threadIdx.x < TP && tile0 + (int)threadIdx.x < a.p) {
    klc_first = a.ell_klc[tile0 + (int)threadIdx.x];
    klc_loaded = true;
  }
  float rel_shift = 0.f;
  if (f_prev) {  // base.py:324: tol * mean(H) of the state being evaluated (global row sums)
    double tot = 0.0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) tot += a.hstat_in[ESPM_HS_ROWSUM + kk];
    rel_shift = (float)((double)a.rel_tol * tot * a.inv_count);
  }
  float mhv[K];   // GLOBAL max over pixels of every row of H, updates.py:139
#pragma unroll
  for (int kk = 0; kk < K; ++kk) mhv[kk] = (RULE == 0 && f_lap) ? (float)a.hstat_in[ESPM_HS_MAX + kk] : 0.f;
  __syncthreads();
  ESPM_PHASE_STAMP(3);

  // ---- epilogue: one thread per pixel -------------------------------------------------------
  auto emit_ht = [&](int q, int jj, const float (&ht)[KP]) {   // the transposed copy of the new column: memory, or the W walk's LDS table
    if (lds_tab) {
      float4 hi = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KP > 4 && K > 4) hi = make_float4(ht[4], ht[5], ht[6], ht[7]);
      FixTab<K>::put(lds_tab, lds_rows, jj, make_float4(ht[0], ht[1], ht[2], ht[3]), hi);   // (a table handed to the epilogue is the fused kernel's)
    } else {
      store_row_kp(a.h_t + (size_t)q * KP, ht);
    }
  };
  const float ls = a.lambda_l * a.sigma_l;
  for (int jj = threadIdx.x; jj < TP; jj += (int)blockDim.x) {
    const int q = tile0 + jj;
    if (q >= a.p) {
      if (lds_tab) FixTab<K>::put(lds_tab, lds_rows, jj, make_float4(1.f, 1.f, 1.f, 1.f), make_float4(1.f, 1.f, 1.f, 1.f));
      continue;
    }
    if (!loaded) h_epilogue_load<K, PLAIN>(a, q, stencil, in);  // tiles wider than the workgroup: later pixels of a thread
    loaded = false;
    const float klc = klc_loaded ? klc_first : (want_klc ? a.ell_klc[q] : 0.f);
    klc_loaded = false;
    float hin[K], nv[K], dv[K];
    const int prows = kl_rows ? K + 1 : K;   // rows of TP floats per partial
    if constexpr (MAXP > 0) {
      // every read unconditional, from the address of a slot that exists (a conditional LDS load compiles to a branch around it, and the
      // reads then wait for one another: 48 dependent round trips of ~100 cycles were the 2.06 us a 64-row shard's update spent here,
      // 24 of them the 0.82 us at the headline, profiles/r03bh_phase_clock_*rows.log); the values of slots beyond nparts are dropped by a
      // select.  Row by row: MAXP reads in flight, the next row's behind them as far as the registers allow (all (K + 1) MAXP at once
      // spilled 111 registers in the k = 8 instance below the full geometry).
      // ESPM_SUM_ROW_FENCE: the scheduler may not carry the reads of one row over into the next (left alone it requests ALL rows' slots at
      // once: 92 registers spilled to scratch in the k = 8 instance below the full geometry - 30 us of update in a 128-row shard of
      // configuration 5 instead of 5, profiles/r04af_phase_c5_128.log).  ESPM_SUM_SWITCH: as many reads per row as the launch has slots
      // (2, 4 or MAXP; uniform branch) instead of MAXP reads of which nparts count.
      auto row_sum_n = [&](auto nslots, int kk, float s) {
        constexpr int NS = decltype(nslots)::value;
        float pv[NS];
#pragma unroll
        for (int w = 0; w < NS; ++w) pv[w] = smem[((size_t)(w < nparts ? w : 0) * prows + kk) * TP + jj];
#pragma unroll
        for (int w = 0; w < NS; ++w) s += w < nparts ? pv[w] : 0.f;
#if ESPM_SUM_ROW_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
        return s;
      };
      auto sum_rows = [&](auto nslots) {
        if (kl_rows) red[ESPM_HP_KL] += row_sum_n(nslots, K, fmaxf(klc, 0.f));   // (negative: the mark of a pixel without counts, no constant)
        else if (unit_kl) red[ESPM_HP_KL] += fmaxf(klc, 0.f);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) nv[kk] = row_sum_n(nslots, kk, 0.f) * a.xscale;
      };
      if (ESPM_SUM_SWITCH && MAXP > 2 && nparts <= 2) sum_rows(std::integral_constant<int, 2>());
      else if (ESPM_SUM_SWITCH && MAXP > 4 && nparts <= 4) sum_rows(std::integral_constant<int, 4>());
      else sum_rows(std::integral_constant<int, MAXP>());
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        hin[kk] = in.hin[kk];
        dv[kk] = (float)(colsum ? colsum[kk] : a.colsum_gw[kk]);
      }
    } else {
      if (kl_rows) {
        float s = fmaxf(klc, 0.f);   // (negative: the mark of a pixel without counts, no constant)
        for (int w = 0; w < nparts; ++w) s += smem[((size_t)w * prows + K) * TP + jj];
        red[ESPM_HP_KL] += s;
      } else if (unit_kl) {
        red[ESPM_HP_KL] += fmaxf(klc, 0.f);
      }
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        float s = 0.f;
        for (int w = 0; w < nparts; ++w) s += smem[((size_t)w * prows + kk) * TP + jj];
        hin[kk] = in.hin[kk];
        nv[kk] = s * a.xscale;
        dv[kk] = (float)(colsum ? colsum[kk] : a.colsum_gw[kk]);
      }
    }
    ESPM_PHASE_STAMP(40);   // (instrumented build) partial numerators summed
    if (f_fill) {  // (uniform) sparse store: a pixel without counts takes the numerator of its log_shift fill (include/espm_mu.h)
      const float mark = klc;
      if (mark < 0.f) {
        const int idx = (int)(-mark) - 1;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) nv[kk] += a.fill_num[(size_t)kk * a.fill_n + idx] * a.xscale;
      }
    }
    if (f_breg) {  // Bregman variant, updates.py:120-125: num = sR / H, denum = colsum(GW) - GW^T (X / GWH) + sR / H
      const float sr = a.xscale * a.breg_sr[q];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float t = sr * __builtin_amdgcn_rcpf(hin[kk]);
        dv[kk] = (dv[kk] - nv[kk]) + t;
        nv[kk] = t;
      }
    }
    if (f_l2) {  // Frobenius branch: denum = (GW^T GW) H, updates.py:115-118
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        float d = 0.f;
#pragma unroll
        for (int l = 0; l < K; ++l) d = fmaf(a.l2_m[kk * KP + l], hin[l], d);
        dv[kk] = d;
      }
    }
    if (f_prev) {
      // rel_H of the update that produced h_in: the other buffer still holds the previous H
      // (each thread reads its own entries before overwriting them below), base.py:324
      float worst = 0.f;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) worst = fmaxf(worst, fabsf(hin[kk] - in.hprev[kk]) * __builtin_amdgcn_rcpf(hin[kk] + rel_shift));   // (a stop-rule statistic: 1 ulp reciprocal)
      red[R_RELH] = fmaxf(red[R_RELH], worst);
    }
    constexpr bool quad = RULE == 1;   // quadratic surrogate of the Laplacian term (multiplicative_step_hq, updates.py:263-315)
    constexpr bool pgrad = RULE == 2;  // projected gradient (proj_grad_step_h, updates.py:372-395)
    if (f_mu) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float m = a.mu[kk];
        if (!quad) dv[kk] += m * __builtin_amdgcn_rcpf(hin[kk] + a.eps_reg);   // updates.py:134-137 (mu is not in the hq update)
        red[ESPM_HP_REG] += m * logf(hin[kk] + a.eps_reg);  // measures.py:543-548
      }
    }
    float hlv[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) hlv[kk] = 0.f;
    if (f_lap) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        // (H L)[q] = deg(q) H[q] - sum of the existing neighbours, utils.py:39-76
        const float hl = f_grid ? ((in.wl + in.wr) + (in.wu + in.wd)) * hin[kk] -
                                           (((in.wl * in.l[kk] + in.wr * in.r[kk]) + in.wu * in.u[kk]) + in.wd * in.d[kk])
                                     : hin[kk];
        hlv[kk] = hl;
        red[ESPM_HP_LAP] += hin[kk] * hl;             // measures.py:574-577
        if constexpr (RULE == 0) {
          const float mh = mhv[kk];
          nv[kk] += ls * mh;                                      // updates.py:140
          dv[kk] += ls * mh + a.lambda_l * hl;                    // updates.py:141
        }
        if constexpr (pgrad) dv[kk] += a.lambda_l * hl;           // gradient of the Laplacian term, updates.py:349-350
      }
    }
    if (!f_write) continue;
    if constexpr (pgrad) {
      // H - grad / gamma with grad = -GW^T (X / GWH) + colsum(GW) + mu / (H + eps) + lambda (H L) = dv - nv, then the
      // projection on the simplex: nu with sum_k max(h_k + nu, eps) = 1 (dicotomy.py:84-108).  The sum is convex, piecewise
      // linear and increasing: Newton from the right end of the reference's bracket reaches the root in at most K steps.
      const float inv_gamma = __builtin_amdgcn_rcpf(a.sigma_l);   // gamma_H travels in sigma_l
      float hg[K];
      float hmin = INFINITY;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        hg[kk] = hin[kk] - (dv[kk] - nv[kk]) * inv_gamma;
        hmin = fminf(hmin, hg[kk]);
      }
      float nu = 0.f;
      if (f_simplex) {
        nu = 1.f / (float)K - hmin;
        for (int it = 0; it < K + 2; ++it) {
          float f = -1.f, cnt = 0.f;
#pragma unroll
          for (int kk = 0; kk < K; ++kk) {
            const float t = hg[kk] + nu;
            if (t > a.log_shift) { f += t; cnt += 1.f; } else { f += a.log_shift; }
          }
          if (!(cnt > 0.f) || fabsf(f) <= 2e-7f) break;
          nu -= f / cnt;
        }
      }
      float ht[KP];
#pragma unroll
      for (int kk = 0; kk < KP; ++kk) ht[kk] = 0.f;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        float hn = fmaxf(hg[kk] + nu, a.log_shift);
        if (f_fixed) {
          const float f = a.fixed_h[(size_t)kk * a.p_pad + q];
          if (f >= 0.f) hn = f;
        }
        {
          const double dh = (double)hn - (double)hin[kk];
          pg_q += dh * (double)(dv[kk] - nv[kk]) + (double)a.sigma_l * dh * dh;
        }
        if (!(hn <= 3.0e38f)) red[ESPM_HP_BAD] += 1.f;
        a.h_out[(size_t)kk * a.p_pad + q] = hn;
        ht[kk] = hn;
        red[R_ROWSUM + kk] += hn;
        red[R_MAX + kk] = fmaxf(red[R_MAX + kk], hn);
      }
      emit_ht(q, jj, ht);
      continue;
    }
    if constexpr (quad) if (f_lap) {
      // a H'^2 + b H' - c = 0 with a = lambda sigma, b = colsum(GW) + lambda (H L) - lambda sigma H (+ nu), c = H GW^T (X / GWH)
      float bq[K], cq[K];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        cq[kk] = hin[kk] * nv[kk];
        bq[kk] = dv[kk] + a.lambda_l * hlv[kk] - ls * hin[kk];
      }
      const float inv2a = __builtin_amdgcn_rcpf(2.f * ls);
      float nu = 0.f;
      if (f_simplex && !simplex_root_hq<K>(ls, bq, cq, a.log_shift, 100, nu)) red[ESPM_HP_BAD] += 1.f;
      float ht[KP];
#pragma unroll
      for (int kk = 0; kk < KP; ++kk) ht[kk] = 0.f;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        float hn = fmaxf(hq_root(bq[kk] + nu, 4.f * ls * cq[kk]) * inv2a, a.log_shift);   // updates.py:300, :307
        if (f_fixed) {
          const float f = a.fixed_h[(size_t)kk * a.p_pad + q];
          if (f >= 0.f) hn = f;
        }
        if (!(hn <= 3.0e38f)) red[ESPM_HP_BAD] += 1.f;
        a.h_out[(size_t)kk * a.p_pad + q] = hn;
        ht[kk] = hn;
        red[R_ROWSUM + kk] += hn;
        red[R_MAX + kk] = fmaxf(red[R_MAX + kk], hn);
      }
      emit_ht(q, jj, ht);
      continue;
    }
    ESPM_PHASE_STAMP(41);   // regularisers, stencil (the pixel's loads have arrived)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) nv[kk] *= hin[kk];          // updates.py:142
#ifdef ESPM_EXPERIMENT_NO_SIMPLEX_ROOT   // TIMING ONLY: what the per-pixel multiplier search costs on the critical path
    if (false) {
#else
    if (f_simplex) {
#endif
      float delta, e[K];
      if (!simplex_root<float, K>(nv, dv, K, a.log_shift, fminf(a.tol, 1e-6f), 100, delta, e, ESPM_SIMPLEX_FAST_EXIT != 0)) red[ESPM_HP_BAD] += 1.f;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) dv[kk] = e[kk] + delta;  // = den + nu, formed without cancellation
    }
    ESPM_PHASE_STAMP(42);   // simplex multiplier found
    float ht[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) ht[kk] = 0.f;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      float hn = fmaxf(nv[kk] * __builtin_amdgcn_rcpf(dv[kk]), a.log_shift);   // updates.py:152 (v_rcp_f32: 1 ulp)
      if (f_fixed) {
        const float f = a.fixed_h[(size_t)kk * a.p_pad + q];
        if (f >= 0.f) hn = f;                                   // updates.py:154-155
      }
      if (!(hn <= 3.0e38f)) red[ESPM_HP_BAD] += 1.f;            // NaN or inf
      // fused half-steps: H' written through (sc1) - 4 k p bytes that would otherwise sit dirty in the XCDs' L2s until the launch ends and
      // lengthen the boundary behind it (MI355X_MICROARCH.md: + B / 6 TB/s); nobody in this launch reads them.  Headline 135.3 -> 133.3 us per
      // iteration, same bits (profiles/r04aa_hout_write_through_ab_512.log)
      if (ESPM_HOUT_WRITE_THROUGH && lds_tab) __hip_atomic_store(a.h_out + (size_t)kk * a.p_pad + q, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.h_out[(size_t)kk * a.p_pad + q] = hn;
      ht[kk] = hn;
      red[R_ROWSUM + kk] += hn;
      red[R_MAX + kk] = fmaxf(red[R_MAX + kk], hn);
    }
    emit_ht(q, jj, ht);
  }

  if (unit_kl && (int)threadIdx.x < n_unit_kl) red[ESPM_HP_KL] += unit_kl[threadIdx.x];   // (thread u: unit u's sum - a fixed assignment, whoever walked the unit)
  ESPM_PHASE_STAMP(4);
  after_pixels();
  // field-major records: hpart[field][block], so that the finalize kernel reads them coalesced; thread i of the
  // workgroup finishes and writes value i itself
  const size_t nb = a.rec_nb ? (size_t)a.rec_nb : gridDim.x - a.tail_on;   // (an extra workgroup may carry the previous W update's tail: not a record)
  double* out = a.hpart + (a.rec_nb ? 2 * blockIdx.x : blockIdx.x);
  auto emit = [&](int i, double v) {
    // red[]: [0..3] KL, REG, LAP, BAD, [4..4+K) row sums, [4+K] RELH, [5+K..5+2K) maxima -> record fields
    const int field = i < R_RELH ? i : (i == R_RELH ? ESPM_HP_RELH : (i == R_RELW ? ESPM_HP_RELW : ESPM_HP_MAX + (i - R_MAX)));
    out[(size_t)field * nb] = v;
  };
  if (red_scratch && RULE != 2) {
    // a wave reduces its own values as soon as its pixels are done (while slower waves are still at theirs) into scratch nobody
    // else touches; the one barrier that follows is also the one the W walk needs (H' table complete); 15 threads then add the
    // waves' values and store the record while everybody else is already walking
    if constexpr (ESPM_FUSED_RED_PACKED && R_RELH <= 16 && NRED - R_RELH <= 16)
      block_reduce_f32_wave_packed<NRED, R_RELH>(red, red_scratch);
    else   // (more than 12 components: the wide build's fused instances)
      block_reduce_f32_wave<NRED, R_RELH>(red, red_scratch);
    __syncthreads();
    block_reduce_f32_finish<NRED, R_RELH>(red_scratch, emit);
  } else {
    __syncthreads();  // smem is reused as reduction scratch
    block_reduce_f32<NRED, R_RELH>(red, reinterpret_cast<double*>(smem), emit);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + 2 * (KP - K)) {   // the unused component slots of the record: zeros (second wave: off the reducing lanes' path)
    const int j = threadIdx.x - 64;
    out[(size_t)((j < KP - K ? ESPM_HP_ROWSUM + K + j : ESPM_HP_MAX + K + (j - (KP - K)))) * nb] = 0.0;
  }
  if (a.rec_nb && 2 * blockIdx.x + 1 < nb && threadIdx.x >= 128 && threadIdx.x <= 128 + ESPM_HP_RELH)   // the slot of the block's second tile: sums + 0, maxima of non-negative values with 0
    out[(size_t)(threadIdx.x - 128) * nb + 1] = 0.0;
  if (a.rec_nb && 2 * blockIdx.x + 1 < nb && threadIdx.x == 192) out[(size_t)ESPM_HP_RELW * nb + 1] = -1.0;   // (and "no share of rel_W")
  if constexpr (RULE == 2) {
    __syncthreads();
    double one[1] = {pg_q};
    block_reduce<1, 1>(one, reinterpret_cast<double*>(smem));
    if (threadIdx.x == 0) a.hpart[(size_t)ESPM_HP_PGQ * nb + blockIdx.x] = one[0];
  }
}

// K components, XT storage type of X, PX pixels per lane (tile = 64 * PX pixels), NW waves per
// workgroup (they split the channel range), LOSS: accumulate the KL term, U channels per load group,
// NBUF: depth of the register ring of X load groups kept in flight (0 / 1: no explicit prefetch).
// L2: the Frobenius branch (updates.py:109-118): num = GW^T X, no ratio, no loss; the epilogue takes the denominator
// (GW^T GW) H from a.l2_m.
template <int K, typename XT, int PX, int NW, bool LOSS, int U, int NBUF, bool L2 = false, int RULE = 0>
__global__ __launch_bounds__(NW * 64, (K > 12 && NW == 4) ? 2 : 1) void h_step_kernel(const HStepArgs a) {   // (k > 12: two workgroups per CU, <= 256 registers)
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][K][TP]
  constexpr int TP = 64 * PX;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * TP;
  const int px0 = tile0 + lane * PX;

  // pixel PAIRS in fp32x2 registers: with a wave-uniform (SGPR) GW operand a plain v_fma_f32 issues at
  // half rate on gfx950, v_pk_fma_f32 with the scalar broadcast does two FMAs in the same slot
  // (tools/ubench/valu_rate.hip: 4.4 vs 2.3 cycles per FMA)
  constexpr int P2 = PX / 2;
  static_assert(PX % 2 == 0, "pixels per lane must be even");
  f2 h[K][P2];
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    float t[PX];
    load_f32<PX>(a.h_in + (size_t)kk * a.p_pad + px0, t);
#pragma unroll
    for (int i = 0; i < P2; ++i) h[kk][i] = f2{t[2 * i], t[2 * i + 1]};
  }

  f2 num[K][P2];
  f2 kl[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    kl[i] = f2{0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < K; ++kk) num[kk][i] = f2{0.f, 0.f};
  }

  const int chunk = (a.n + NW - 1) / NW;
  const int c_begin = min(a.n, wave * chunk);
  const int c_end = min(a.n, c_begin + chunk);
  // X is tile-major: x_cm[pixel block of x_tile][channel][x_tile]; a workgroup streams one contiguous
  // region of it (sequential DRAM pages, few TLB entries) with row stride x_tile
  const XT* xbase = static_cast<const XT*>(a.x_cm) + (size_t)(tile0 / a.x_tile) * a.n_cm * a.x_tile + (tile0 % a.x_tile) +
                    lane * PX;

  // one channel: Y = GW[c,:] H, R = X / Y, num += GW[c,:]^T R   (updates.py:127-128)
  auto channel = [&](const XVec<XT, PX>& xv, const float (&gk)[K]) {
    f2 x[P2];
    xv.get2(x);
#pragma unroll
    for (int i = 0; i < P2; ++i) {
      f2 y = gk[0] * h[0][i];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = gk[kk] * h[kk][i] + y;
      const f2 inv = f2{__builtin_amdgcn_rcpf(y.x), __builtin_amdgcn_rcpf(y.y)};
      // R = X / Y; with the loss the tiny offset keeps log2(R) finite where X = 0 (0 * finite = 0) at
      // no extra cost (it rides in the fma) and is far below fp32 resolution of any non-zero R
      const f2 r = L2 ? x[i] : (LOSS ? x[i] * inv + f2{1e-37f, 1e-37f} : x[i] * inv);
#pragma unroll
      for (int kk = 0; kk < K; ++kk) num[kk][i] = gk[kk] * r + num[kk][i];
      if constexpr (LOSS) kl[i] = x[i] * f2{__builtin_amdgcn_logf(r.x), __builtin_amdgcn_logf(r.y)} + kl[i];
    }
  };

  // Loads are issued NBUF groups of U channels ahead of their use (register ring, static indices after
  // unrolling); the wave-uniform GW rows (scalar cache) one group ahead.  Prefetch addresses are clamped
  // to the last full group of the wave's chunk, so no load is predicated and none leaves the chunk (a
  // clamped re-load is simply not consumed).
  struct XGroup {
    XVec<XT, PX> x[U];
  };
  struct GGroup {
    float g[U][K];
  };
  auto load_x = [&](XGroup& grp, int c) {
    const XT* xr = xbase + (size_t)c * a.x_tile;
#pragma unroll
    for (int u = 0; u < U; ++u) grp.x[u].load(xr + (size_t)u * a.x_tile);
  };
  auto load_g = [&](GGroup& grp, int c) {
    const float* gr = a.gw_s + (size_t)c * KP;  // wave-uniform -> scalar loads
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) grp.g[u][kk] = gr[u * KP + kk];
  };
  auto compute_group = [&](const XGroup& xg, const GGroup& gg) {
#pragma unroll
    for (int u = 0; u < U; ++u) channel(xg.x[u], gg.g[u]);
  };

  int c = c_begin;
  if constexpr (NBUF >= 2) {
    const int ngroups = (c_end - c_begin) / U;
    if (ngroups > 0) {
      const int c_last = c_begin + (ngroups - 1) * U;
      XGroup xs[NBUF];
      GGroup ga, gb;
#pragma unroll
      for (int b = 0; b < NBUF; ++b) load_x(xs[b], min(c + b * U, c_last));
      load_g(ga, c);
      int g = 0;
      // two passes over the ring per trip so that the GW double buffer (ga / gb) keeps static roles
      for (; g + 2 * NBUF <= ngroups; g += 2 * NBUF) {
#pragma unroll
        for (int b = 0; b < 2 * NBUF; ++b) {
          GGroup& cur = (b & 1) ? gb : ga;
          GGroup& nxt = (b & 1) ? ga : gb;
          load_g(nxt, min(c + U, c_last));
          compute_group(xs[b % NBUF], cur);
          load_x(xs[b % NBUF], min(c + NBUF * U, c_last));
          c += U;
        }
      }
      // remaining full groups: their X is already in the ring (slot (g + b) % NBUF == b since g is a
      // multiple of NBUF); beyond the ring fall through to the plain loop below
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        if (g < ngroups) {
          GGroup gl;
          load_g(gl, c);
          compute_group(xs[b], gl);
          c += U;
          ++g;
        }
      }
    }
  }
  for (; c + U <= c_end; c += U) {
    XGroup xg;
    GGroup gg;
    load_x(xg, c);
    load_g(gg, c);
    compute_group(xg, gg);
  }
  for (; c < c_end; ++c) {  // remainder channels one at a time
    XVec<XT, PX> xv;
    xv.load(xbase + (size_t)c * a.x_tile);
    float gk[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) gk[kk] = a.gw_s[(size_t)c * KP + kk];
    channel(xv, gk);
  }

  // ---- cross-wave reduction of the numerators through LDS ---------------------------------
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    float* dst = smem + ((size_t)wave * K + kk) * TP + lane * PX;
#pragma unroll
    for (int i = 0; i < P2; ++i) {
      dst[2 * i] = num[kk][i].x;
      dst[2 * i + 1] = num[kk][i].y;
    }
  }
  float kl_lane = 0.f;
  if constexpr (LOSS) {
#pragma unroll
    for (int i = 0; i < P2; ++i) kl_lane += kl[i].x + kl[i].y;
  }
  h_epilogue<K, true, RULE>(a, smem, NW, TP, tile0, kl_lane);
}

}  // namespace espm
