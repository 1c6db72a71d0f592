// Kernels of the sparse count store (x_dtype = ESPM_X_ELL, include/espm_mu.h).
//
// Spectrum images are Poisson counts: at the headline size (500 counts per pixel over 2048 channels) 80 % of
// the entries of X are zero, and a zero entry contributes nothing to either update (R = X / Y is zero there,
// and X ln(X / Y) too) - the dense kernels spend four fifths of their vector-ALU time, which is what bounds
// them, on those zeros.  Here only the non-zero entries are stored and processed, 16 bits each:
//
//   H-step: lists BY PIXEL.  A lane owns one pixel, so H[:, pixel] and the numerator of that pixel live in
//           registers and nothing is reduced across lanes; per entry the lane gathers the GW row of the
//           entry's channel from a table in LDS.  entry = count << cbits | channel.
//   W-step: lists BY CHANNEL inside blocks of 1024 pixels.  A lane owns one channel (GW row and the R H^T
//           accumulator in registers) and gathers H[:, pixel] of the entry from a table of the block's
//           H columns in LDS.  entry = count << pbits | pixel-in-block.
//   In both sets the 64 lists of a wave are neighbours in the order of decreasing list length (pixels inside
//   the workgroup's window: pix_perm; channels inside the pixel block: chan_perm), so they have about the same
//   length and the padding stays at a few per cent.
//
// Both lists are "ELL" slabs: the j-th entries of the 64 lanes of a wave are adjacent in memory (one
// coalesced 256-byte row carries entries 2r and 2r+1 of each lane), padded with zero entries (count 0) to the
// longest list of the 64.  A count that does not fit its field is stored as several entries x_i of the same
// channel / pixel; the loss term is accumulated per entry as x_i log2(x_i / Y) (well conditioned, like the dense
// kernels) and the per-pixel constant  sum x log2 x - sum_i x_i log2 x_i  of the split counts is added at the end.
//
// Nine in ten non-zero entries of a count image are ONES.  The rows of a 64-list group therefore come in two
// segments: first the UNIT rows - every lane holds an entry with count 1 in every position, stored as the byte
// offset of its table row (index << 4), no count field, no padding: two vector instructions decode an entry,
// x = 1 drops the multiply and the padding guard - then the general rows described above, which take the rest
// (counts > 1, the ones beyond the shortest run of ones among the 64 lanes, padding).  The offsets array has
// two words per group: [first unit row, first general row], and the end of the last group.
//
// Bounds: the lists are read once per launch at HBM rate; per entry the LDS serves one 16- or
// 20..32-byte gather, which is the second limit (random rows: ~3-way bank conflicts inside a 16-lane group).
#pragma once
#include "mu_h_kernel.hpp"


namespace espm {

// LDS table of rows of K floats: components 0..3 as float4 (ds_read_b128 at a 16-byte stride: the 16 lanes of
// a read group spread over all 16 bank quads), components 4.. in a second array of 1, 2 or 4 floats per row.
// The second array comes FIRST: with the table at the start of the workgroup's LDS a unit entry (byte offset of
// the float4 row) addresses it as entry >> 2 (or >> 1, >> 0) without adding a base.
typedef float lds_v4f __attribute__((ext_vector_type(4)));
typedef float lds_v2f __attribute__((ext_vector_type(2)));
#define ESPM_LDS(T) __attribute__((address_space(3))) const T*
__device__ __forceinline__ uint32_t lds_address(const float* ptr) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)ptr;
}
// The kernels below have no static LDS, so their dynamic LDS (the table) starts at address 0; the unit-entry
// decode relies on it and every kernel checks it once.
__device__ __forceinline__ void ell_table_at_lds_zero(const float* tab) {
  if (lds_address(tab) != 0u) __builtin_trap();
}
template <int K>
struct EllTab {
  static constexpr int WB = LdsTabGeom<K>::WB;   // (mu_h_kernel.hpp) 0, 1, 2, 4 | 8 | 12 floats beyond the float4 part
  static constexpr int FLOATS = 4 + WB;
  static constexpr int WQ = WB / 4;              // float4 of the second array per row (from 7 components on)
  static __device__ __forceinline__ const float* quad(const float* tab, int rows) { return tab + (size_t)WB * rows; }
  // row r from a KP-strided row of gw_s / h_t
  static __device__ __forceinline__ void put(float* tab, int rows, int r, const float* src) {
    float row[(K + 3) / 4 * 4];   // (16-byte loads of the quads that hold a component)
#pragma unroll
    for (int i = 0; i < (K + 3) / 4 * 4; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      row[i] = v.x; row[i + 1] = v.y; row[i + 2] = v.z; row[i + 3] = v.w;
    }
    lds_table_put_row<K>(tab, rows, r, row);
  }
  static __device__ __forceinline__ void get(const float* tab, int rows, uint32_t r, float (&g)[K]) {
#ifdef ESPM_EXPERIMENT_GENERAL_NOCONFLICT   // TIMING ONLY (wrong results): what the general rows would cost if their gathers never met in a bank
    r = (threadIdx.x & 15) + 16 * ((r >> 4) & 7);
#endif
    const float4 lo = reinterpret_cast<const float4*>(quad(tab, rows))[r];
    const float l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
    for (int i = 0; i < (K < 4 ? K : 4); ++i) g[i] = l[i];
    if constexpr (WB == 1) g[4] = tab[r];
    if constexpr (WB == 2) {
      const float2 v = reinterpret_cast<const float2*>(tab)[r];
      g[4] = v.x;
      g[5] = v.y;
    }
    if constexpr (WB >= 4) {
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const float4 v = reinterpret_cast<const float4*>(tab)[(size_t)r * WQ + q];
        const float h[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 + 4 * q + i < K) g[4 + 4 * q + i] = h[i];
      }
    }
  }
  // unit entry: off = 16 * row.  The table sits at LDS address 0 (ell_table_at_lds_zero), lds_q = LDS address of its
  // float4 part: the address of the second array is a shift (or a small multiple) of the entry, with no base to add.
  static __device__ __forceinline__ void get_unit(uint32_t lds_q, uint32_t off, float (&g)[K]) {
    const lds_v4f lo = *(ESPM_LDS(lds_v4f))(uintptr_t)(lds_q + off);
#pragma unroll
    for (int i = 0; i < (K < 4 ? K : 4); ++i) g[i] = lo[i];
    if constexpr (WB == 1) g[4] = *(ESPM_LDS(float))(uintptr_t)(off >> 2);
    if constexpr (WB == 2) {
      const lds_v2f v = *(ESPM_LDS(lds_v2f))(uintptr_t)(off >> 1);
      g[4] = v[0];
      g[5] = v[1];
    }
    if constexpr (WB >= 4) {
      const uint32_t base = off * WQ;   // (x 1, 2: a shift, or x 3)
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const lds_v4f v = *(ESPM_LDS(lds_v4f))(uintptr_t)(base + 16 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 + 4 * q + i < K) g[4 + 4 * q + i] = v[i];
      }
    }
  }
};

// y = sum_k g[k] h[k] and acc[k] += g[k] r with component PAIRS in fp32x2 registers (v_pk_fma_f32: the float4
// of a gathered row already sits in two aligned register pairs), an odd last component as a scalar FMA.
// ESPM_ELL_DOT_PLAIN: y as a chain of K scalar FMAs instead of K / 2 packed ones + the horizontal add (by the issue costs of
// tools/ubench/valu_rate.hip 12.5 against 14.4 cycles at k = 5).  Measured: no gain (147.9 against 146.9 us per iteration at
// the headline, profiles/r03a_variant_ab_512.log) - off.
#ifndef ESPM_ELL_DOT_PLAIN
#define ESPM_ELL_DOT_PLAIN 0
#endif
// Round 5, measured again on the final kernels (profiles/r05q_ab_*.log, seven interleaved repetitions on one box): at k = 5 the chain of five
// FMAs wins in EVERY repetition - headline 132.8-134.7 -> 131.4-132.4 us per iteration, a 64-row shard 31.4 -> 31.1 - as the issue costs say
// (v_pk_fma_f32 4.0 cycles, v_fma_f32 2.5: 12.6 against 13.4 for an odd k = 5, where the packed form pays a horizontal add AND a scalar FMA);
// k = 3: 108.5 -> 108.9 (no gain), even k: the packed form is cheaper by the same arithmetic.  Hence for k = 5 only.
#ifndef ESPM_ELL_DOT_PLAIN_K
#define ESPM_ELL_DOT_PLAIN_K 5
#endif
template <int K>
__device__ __forceinline__ float ell_dot(const float (&g)[K], const float (&h)[K]) {
  if constexpr (ESPM_ELL_DOT_PLAIN || K == ESPM_ELL_DOT_PLAIN_K) {
    float y = g[0] * h[0];
#pragma unroll
    for (int i = 1; i < K; ++i) y = fmaf(g[i], h[i], y);
    return y;
  } else if constexpr (K == 1) {
    return g[0] * h[0];
  } else {
    f2 s = f2{g[0], g[1]} * f2{h[0], h[1]};
#pragma unroll
    for (int q = 1; q < K / 2; ++q) s = f2{g[2 * q], g[2 * q + 1]} * f2{h[2 * q], h[2 * q + 1]} + s;
    float y = s.x + s.y;
    if constexpr (K & 1) y = fmaf(g[K - 1], h[K - 1], y);
    return y;
  }
}
template <int K>
__device__ __forceinline__ void ell_axpy(float (&acc)[K], const float (&g)[K], float r) {
#pragma unroll
  for (int q = 0; q < K / 2; ++q) {
    const f2 t = f2{g[2 * q], g[2 * q + 1]} * r + f2{acc[2 * q], acc[2 * q + 1]};
    acc[2 * q] = t.x;
    acc[2 * q + 1] = t.y;
  }
  if constexpr (K & 1) acc[K - 1] = fmaf(g[K - 1], r, acc[K - 1]);
}

// Walks the `len` dwords (two 16-bit entries each) of this lane's list; `row` points at the lane's first
// dword, consecutive dwords are 64 apart.  UNR dwords are requested one batch ahead of their use and the
// 2 UNR table gathers of a batch are issued together.  get(dword, half, g) gathers the table row of entry
// `half` of the dword into g and returns the entry's count; body(count, g) consumes one entry.
// flush(redo): called after every batch of 2 UNR entries (and after every dword of the remainder); redo(alt) walks the
// entries of that batch once more with the body `alt` (the logarithm of a product of ratios falls back on it, ell_h_rows).
struct EllNoFlush {
  template <typename Redo>
  __device__ __forceinline__ void operator()(Redo) const {}
};
// PRIO > 0: before every batch the wave sets its issue priority to min(3, dwords left / PRIO).  The waves of a SIMD are served
// oldest first: where every wave walks ONE unit of about equal length (the fused kernel below its full
// geometry) the youngest wave of a SIMD ends long after the oldest, and the SIMD runs one wave deep meanwhile; with the
// priority following the work that is left, whoever is behind is served first.
template <int PRIO>
__device__ __forceinline__ void ell_walk_prio(int left) {
  if constexpr (PRIO > 0) {
    if (left > 3 * PRIO) __builtin_amdgcn_s_setprio(3);
    else if (left > 2 * PRIO) __builtin_amdgcn_s_setprio(2);
    else if (left > PRIO) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
}
// STREAM: the list rows with non-temporal loads.  They are read once per launch; where a launch's lists exceed the 256 MB of the
// last-level cache nothing of them survives until the next launch anyway, and loads that do not allocate leave the cache to the
// data that is re-read (H, the records): -1.9 us of 135 at the headline, -10 of 672 at configuration 5; on a shard whose lists
// FIT the cache the same loads cost +5 us of 32 (profiles/r04ac_*) - espm_mu_state.ell_stream, the caller's decision.
template <bool STREAM>
__device__ __forceinline__ uint32_t ell_list_load(const uint32_t* p) {
  if constexpr (STREAM) return __builtin_nontemporal_load(p);
  else return *p;
}
// ell_walk_pre: `use_pre` (wave-uniform) - the first PF batches were requested by the caller ahead of time (ell_walk_request: the fused
// kernel asks for a wave's first rows while its workgroup is still staging the table, so that the walk starts without a round trip to
// memory of its own) and arrive in `pre`.
template <int UNR, int PF, bool STREAM>
__device__ __forceinline__ bool ell_walk_request(const uint32_t* row, int len, uint32_t (&pre)[PF][UNR]) {
  if (len < UNR) return false;
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    const int jd = min(d * UNR, len - UNR);
#pragma unroll
    for (int u = 0; u < UNR; ++u) pre[d][u] = ell_list_load<STREAM>(row + (size_t)(jd + u) * 64);
  }
  return true;
}
// PP: the list dwords of the batch in hand and of the batch requested ahead in two register sets used in turn (below).  The caller's choice: it
// pays where the registers are there - the fused kernel at its full geometry up to ESPM_ELL_WALK_PP_MAX_K components.  MEASURED at the headline
// image (profiles/r05u_ab_k*_512.log, ring | two sets, us per iteration): k = 1 106.6 | 100.2, k = 2 103.9 | 92.5, k = 3 107.1 | 99.5, k = 4 113.0 | 107.6,
// k = 5 128.4 | 127.2 (and 128.2 | 127.1 in r05t); k = 6 137.6 | 191.6, k = 7 153.5 | 551.7, k = 8 162.3 | 544.6: from 6 components on the second set spills.
// Below the full geometry (a 64-row shard, k = 5) 30.9 | 31.4: not there.  Same rows in the same order: the same bits.
#ifndef ESPM_ELL_WALK_PP_MAX_K
#define ESPM_ELL_WALK_PP_MAX_K 5
#endif
// MEASURED (profiles/r05w_ab_*.log, whole batch | halves, us per iteration at the headline image): k = 6 138.9 | 138.8, k = 7 153.6 | 151.3, k = 8 161.7 | 159.7;
// configuration 5 634.4 | 628.7, its 128-row shard 91.6 | 91.1.  (The two register sets of PP on top still spill at k >= 6: 242 ... 619 us.)  From 7 components on.
#ifndef ESPM_ELL_BATCH_HALVES_K
#define ESPM_ELL_BATCH_HALVES_K 7
#endif
template <int K, int UNR, int PF = 1, int PRIO = 0, bool STREAM = false, bool PP = false, typename Get, typename Body, typename Flush = EllNoFlush>
__device__ __forceinline__ void ell_walk_pre(const uint32_t* row, int len, Get get, Body body, Flush flush, bool use_pre, const uint32_t (&pre)[PF][UNR]) {
  // PF: batches requested ahead of their use (1: the next one - enough with four waves per SIMD taking turns; a workgroup
  // that has a SIMD almost to itself needs the memory latency covered by its own requests)
  auto batch = [&](const uint32_t (&e)[UNR]) {
    // ESPM_ELL_BATCH_HALVES_K: from that many components on the batch's 2 UNR gathers are issued and consumed in two halves - half the
    // gathered rows live at a time (2 UNR K registers are 64 of a wave's 128 at k = 8) - with a scheduling fence between the halves
    constexpr int HALVES = (K >= ESPM_ELL_BATCH_HALVES_K && K <= 8 && UNR % 2 == 0) ? 2 : 1;   // (measured for 7, 8; the wide build's counts keep their batches of 2 dwords whole)
    constexpr int UH = UNR / HALVES;
#pragma unroll
    for (int hh = 0; hh < HALVES; ++hh) {
      float g[2 * UH][K], x[2 * UH];
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        x[2 * u] = get(e[hh * UH + u], 0, g[2 * u]);
        x[2 * u + 1] = get(e[hh * UH + u], 1, g[2 * u + 1]);
      }
#pragma unroll
      for (int u = 0; u < 2 * UH; ++u) body(x[u], g[u]);
      if constexpr (HALVES > 1) __builtin_amdgcn_sched_barrier(0);
    }
    flush([&](auto alt) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float g0[K], g1[K];
        const float x0 = get(e[u], 0, g0), x1 = get(e[u], 1, g1);
        alt(x0, g0);
        alt(x1, g1);
      }
    });
  };
  int j = 0;
  if constexpr (PF == 1 && PP) {
    // One batch requested ahead, in TWO register sets used in turn: the ring below copies the arrived dwords out of the registers the next
    // request is about to overwrite - UNR vector moves per batch, 4 of a k = 5 batch's 102 vector instructions - where two sets need none.
    // The same rows in the same order: results bit for bit.
    if (len >= UNR) {
      uint32_t qa[UNR], qb[UNR];
      if (use_pre) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) qa[u] = pre[0][u];
      } else {
#pragma unroll
        for (int u = 0; u < UNR; ++u) qa[u] = ell_list_load<STREAM>(row + (size_t)u * 64);
      }
      for (; j + 2 * UNR <= len; j += 2 * UNR) {
        ell_walk_prio<PRIO>(len - j);
        const int jb = min(j + UNR, len - UNR);
#pragma unroll
        for (int u = 0; u < UNR; ++u) qb[u] = ell_list_load<STREAM>(row + (size_t)(jb + u) * 64);
        batch(qa);
        ell_walk_prio<PRIO>(len - j - UNR);
        const int ja = min(j + 2 * UNR, len - UNR);   // (the last batches re-request the last rows: no branch, no overrun)
#pragma unroll
        for (int u = 0; u < UNR; ++u) qa[u] = ell_list_load<STREAM>(row + (size_t)(ja + u) * 64);
        batch(qb);
      }
      if (j + UNR <= len) {   // (one whole batch left: its rows are the ones qa holds)
        ell_walk_prio<PRIO>(len - j);
        batch(qa);
        j += UNR;
      }
    }
  } else if (len >= UNR) {
    uint32_t q[PF][UNR];
    if (use_pre) {
#pragma unroll
      for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[d][u] = pre[d][u];
    } else {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int jd = min(d * UNR, len - UNR);
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[d][u] = ell_list_load<STREAM>(row + (size_t)(jd + u) * 64);
      }
    }
    for (; j + UNR <= len; j += UNR) {
      ell_walk_prio<PRIO>(len - j);
      uint32_t e[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) e[u] = q[0][u];
#pragma unroll
      for (int d = 0; d + 1 < PF; ++d)
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[d][u] = q[d + 1][u];
      const int jn = min(j + PF * UNR, len - UNR);  // the last batches re-request the last rows (no branch, no overrun)
#pragma unroll
      for (int u = 0; u < UNR; ++u) q[PF - 1][u] = ell_list_load<STREAM>(row + (size_t)(jn + u) * 64);
      batch(e);
    }
  }
  for (; j < len; ++j) {
    const uint32_t v = ell_list_load<STREAM>(row + (size_t)j * 64);
    float g0[K], g1[K];
    const float x0 = get(v, 0, g0), x1 = get(v, 1, g1);
    body(x0, g0);
    body(x1, g1);
    flush([&](auto alt) {
      float a0[K], a1[K];
      const float y0 = get(v, 0, a0), y1 = get(v, 1, a1);
      alt(y0, a0);
      alt(y1, a1);
    });
  }
}
template <int K, int UNR, int PF = 1, int PRIO = 0, bool STREAM = false, bool PP = false, typename Get, typename Body, typename Flush = EllNoFlush>
__device__ __forceinline__ void ell_walk(const uint32_t* row, int len, Get get, Body body, Flush flush = Flush()) {
  uint32_t none[PF][UNR];   // (never read)
  ell_walk_pre<K, UNR, PF, PRIO, STREAM, PP>(row, len, get, body, flush, false, none);
}
// general entries: count << idx_bits | index
template <int K>
struct EllGet {
  const float* tab;
  int rows, idx_bits;
  uint32_t mask;
  __device__ __forceinline__ EllGet(const float* t, int r, int bits) : tab(t), rows(r), idx_bits(bits), mask((1u << bits) - 1u) {}
  __device__ __forceinline__ float operator()(uint32_t e, int half, float (&g)[K]) const {
    const uint32_t v = half ? e >> 16 : e & 0xffffu;
    EllTab<K>::get(tab, rows, v & mask, g);
    return (float)(v >> idx_bits);
  }
};
// unit entries: byte offset of the float4 row, count 1
template <int K>
struct EllGetUnit {
  uint32_t lds_q;
  __device__ __forceinline__ EllGetUnit(int rows) : lds_q((uint32_t)(EllTab<K>::WB * rows) * 4u) {}
  __device__ __forceinline__ float operator()(uint32_t e, int half, float (&g)[K]) const {
    EllTab<K>::get_unit(lds_q, half ? e >> 16 : e & 0xffffu, g);
    return 1.f;
  }
};

// The fused kernel's tables (mu_h_kernel.hpp: FixTab): one address register per entry, the second part of a row at a constant offset.
template <int K>
struct EllGetUnitFix {
  static constexpr bool FIXED = FixTab<K>::TWO && FixTab<K>::FIXED;
  uint32_t lds_q;   // (only without the fixed layout: the float4 part's run-time base)
  __device__ __forceinline__ EllGetUnitFix(int rows = 0) : lds_q((uint32_t)(EllTab<K>::WB * rows) * 4u) {}
  static __device__ __forceinline__ void row(uint32_t off, float (&g)[K]) {   // off = 16 * row: LDS address of the row's float4
    const lds_v4f lo = *(ESPM_LDS(lds_v4f))(uintptr_t)off;
#pragma unroll
    for (int i = 0; i < (K < 4 ? K : 4); ++i) g[i] = lo[i];
    if constexpr (K == 5) g[4] = *(ESPM_LDS(float))(uintptr_t)(off + ESPM_TAB2_BASE);
    if constexpr (K == 6) {
      const lds_v2f v = *(ESPM_LDS(lds_v2f))(uintptr_t)(off + ESPM_TAB2_BASE);
      g[4] = v[0];
      g[5] = v[1];
    }
    if constexpr (K >= 7) {
      const lds_v4f v = *(ESPM_LDS(lds_v4f))(uintptr_t)(off + ESPM_TAB2_BASE);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 + i < K) g[4 + i] = v[i];
    }
  }
  __device__ __forceinline__ float operator()(uint32_t e, int half, float (&g)[K]) const {
    if constexpr (FIXED || K <= 4) row(half ? e >> 16 : e & 0xffffu, g);
    else EllTab<K>::get_unit(lds_q, half ? e >> 16 : e & 0xffffu, g);
    return 1.f;
  }
};
template <int K>
struct EllGetFix {
  const float* tab;
  int rows, idx_bits;
  uint32_t mask;
  __device__ __forceinline__ EllGetFix(const float* t, int r, int bits) : tab(t), rows(r), idx_bits(bits), mask((1u << bits) - 1u) {}
  __device__ __forceinline__ float operator()(uint32_t e, int half, float (&g)[K]) const {
    const uint32_t v = half ? e >> 16 : e & 0xffffu;
    if constexpr (EllGetUnitFix<K>::FIXED || K <= 4) EllGetUnitFix<K>::row((v & mask) << 4, g);
    else EllTab<K>::get(tab, rows, v & mask, g);
    return (float)(v >> idx_bits);
  }
};
template <int K, bool FIX>
struct EllGetters {
  typedef EllGetUnit<K> Unit;
  typedef EllGet<K> General;
};
template <int K>
struct EllGetters<K, true> {
  typedef EllGetUnitFix<K> Unit;
  typedef EllGetFix<K> General;
};

// The H walk over rows [x0, x1) of a list group whose first `mid` rows are unit rows: num += GW^T (X / (GW H)) of the lane's
// pixel (updates.py:127-128 at the non-zero entries) and, with LOSS, kl += sum x log2(x / y).
// ESPM_ELL_KLPROD: the unit rows' part of the loss, sum log2(1 / y), as ONE logarithm per batch of 2 UNR entries - of the
// product of their ratios (v_log_f32 costs 8 cycles of the vector pipe and the add 2.5, the multiply 2.5).  The product of
// eight ratios can leave the normal range where the single ratios do not (y < ~2e-5 throughout a batch): if ANY lane of the
// wave finds its product not a positive normal number, the wave takes the batch's logarithms entry by entry (the rows are
// gathered once more; wave-uniform branch, not taken on data a fit sees after its first iterations).  The grouping follows the
// rows of the list, not the wave that walks them: the loss stays bit-reproducible.
// Measured at the headline (tools/analysis/variant_ab.py, profiles/r03a_variant_ab_512.log): 146-150 -> 142-148 us per iteration,
// losses equal to 4e-9 relative, W and H bit for bit.
#ifndef ESPM_ELL_KLPROD
#define ESPM_ELL_KLPROD 1
#endif
template <int K, bool LOSS, int UNR, int PF, int PRIO = 0, bool STREAM = false, bool FIX = false, bool PP = false>   // FIX: the fused kernel's table layout; PP: ell_walk_pre
__device__ __forceinline__ void ell_h_rows(const uint32_t* lrow, int x0, int x1, int mid, const float* tab, int n_pad, int ell_bits,
                                           const float (&hk)[K], float (&acc)[K], float& kl) {
  if (x0 < mid) {
    if constexpr (LOSS && ESPM_ELL_KLPROD) {
      float prod = 1.f;
      ell_walk<K, UNR, PF, PRIO, STREAM, PP>(lrow + (size_t)x0 * 64, min(x1, mid) - x0, typename EllGetters<K, FIX>::Unit(n_pad),
        [&](float, const float (&g)[K]) {
          const float r = __builtin_amdgcn_rcpf(ell_dot<K>(g, hk));
          ell_axpy<K>(acc, g, r);
          prod *= r;
        },
        [&](auto redo) {
          const bool ok = __builtin_amdgcn_classf(prod, 0x100);   // a positive normal number
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
            redo([&](float, const float (&g)[K]) { kl += __builtin_amdgcn_logf(__builtin_amdgcn_rcpf(ell_dot<K>(g, hk))); });
          } else {
            kl += __builtin_amdgcn_logf(prod);
          }
          prod = 1.f;
        });
    } else {
      ell_walk<K, UNR, PF, PRIO, STREAM, PP>(lrow + (size_t)x0 * 64, min(x1, mid) - x0, typename EllGetters<K, FIX>::Unit(n_pad), [&](float, const float (&g)[K]) {
        const float r = __builtin_amdgcn_rcpf(ell_dot<K>(g, hk));
        ell_axpy<K>(acc, g, r);
        if constexpr (LOSS) kl += __builtin_amdgcn_logf(r);
      });
    }
  }
  if (x1 > mid) {
    const int g0 = max(x0, mid);
    ell_walk<K, UNR, PF, PRIO, STREAM, PP>(lrow + (size_t)g0 * 64, x1 - g0, typename EllGetters<K, FIX>::General(tab, n_pad, ell_bits), [&](float x, const float (&g)[K]) {
      const float y = ell_dot<K>(g, hk);
      // (+1e-37: a padding entry has x = 0 and must give 0 * log2(tiny), not 0 * -inf; same guard as the dense kernels)
      const float r = LOSS ? fmaf(x, __builtin_amdgcn_rcpf(y), 1e-37f) : x * __builtin_amdgcn_rcpf(y);
      ell_axpy<K>(acc, g, r);
      if constexpr (LOSS) kl = fmaf(x, __builtin_amdgcn_logf(r), kl);
    });
  }
}

// ---- H-step --------------------------------------------------------------------------------------
// One workgroup = 8 waves = TP = 512 / nsplit pixels.  updates.py:127-132 restricted to the non-zero entries of X;
// the per-pixel epilogue (regularisers, simplex, clamp, statistics) is h_epilogue, which sums the partial
// numerators of a pixel.
//   TP = 512 (the headline size), k <= 6: the 8 list groups of the window are in order of decreasing length, so
//     group w is paired with group 7 - w: wave w walks the first half of the pair's rows (all in group w), wave
//     7 - w walks group 7 - w and then the rest of group w.  Every wave carries half a pair - without this the
//     workgroup waits for the wave with the longest lists (+11 % at the headline size) - and a pixel has at
//     most two partial numerators.
//   otherwise every 64-pixel list group is walked by `nsplit` waves, each taking a contiguous slice of its rows
//     (small images use nsplit = 2, 4 or 8 so that the grid still covers the chip).
template <int K, bool LOSS, int UNR, int RULE = 0>
__global__ __launch_bounds__(ESPM_ELL_TILE, (K > 8 ? 2 : 4)) void h_step_ell_kernel(const HStepArgs a) {   // (more than 8 components: 256 registers, one workgroup per CU by its LDS anyway)
  constexpr int NT = ESPM_ELL_TILE;
  constexpr bool PAIRS_OK = K <= ESPM_ELL_PAIR_MAX_K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;                                          // [n_pad] rows of GW
  ell_table_at_lds_zero(tab);
  float* part = smem + (size_t)a.n_pad * EllTab<K>::FLOATS;   // [parts][K][TP] numerators, then reduction scratch
  const int TP = a.ell_tp;             // pixels of this workgroup: 64 * (8 / nsplit)
  const int gpw = TP >> 6;             // list groups per workgroup
  const bool pairs = PAIRS_OK && gpw == NT / 64;
  if (a.tail_on && blockIdx.x == gridDim.x - 1) {   // (uniform) the extra workgroup: tail of the previous W update
    w_tail_body<20>(a.tail, reinterpret_cast<double*>(smem));
    return;
  }
  double* cs_lds = a.cs_parts ? reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(smem) + a.cs_lds_off) : nullptr;
  const bool cs_late = cs_lds && a.cs_lds_off == 0;   // (no room behind the numerators: the sums go where the table was, after the walk - mu_ell.hip)
  auto form_cs = [&]() {   // wave w: column sums w, w + 8, ... of G W' from the W update's partials
    for (int kk = threadIdx.x >> 6; kk < K; kk += NT / 64) {
      double v = 0.0;
      for (int j = threadIdx.x & 63; j < a.cs_nbk; j += 64) v += a.cs_parts[(size_t)kk * a.cs_nbk + j];
      v = wave_sum(v);
      if ((threadIdx.x & 63) == 0) cs_lds[kk] = v;
    }
  };
  if (cs_lds && !cs_late) form_cs();
  for (int r = threadIdx.x; r < a.n_pad; r += NT) EllTab<K>::put(tab, a.n_pad, r, a.gw_s + (size_t)r * KP);
  if (pairs) {  // second partial numerator: only the pixels of the longer group of a pair receive one
#pragma unroll
    for (int kk = 0; kk < K; ++kk) part[((size_t)K + kk) * TP + threadIdx.x] = 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * TP;
  float kl = 0.f;

  // rows [x0, x1) of list group gi of the window -> partial numerator `slot` of its pixels
  auto walk_rows = [&](int gi, int x0, int x1, int slot) {
    const int grp = tile0 / 64 + gi;
    // the lists of a window are ordered by length: slot -> pixel of the window (pad pixels have empty lists)
    const int lp = a.ell_pix[tile0 + gi * 64 + lane];
    const int px = tile0 + lp;  // < p_pad (a multiple of 512)
    float hk[K], acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      hk[kk] = a.h_in[(size_t)kk * a.p_pad + px];
      acc[kk] = 0.f;
    }
    // rows [0, mid) of the group: unit entries, [mid, len): general entries
    const int beg = a.ell_off[2 * grp], mid = a.ell_off[2 * grp + 1] - beg;
    const uint32_t* lrow = a.ell + (size_t)beg * 64 + lane;
    ell_h_rows<K, LOSS, UNR, 1>(lrow, x0, x1, mid, tab, a.n_pad, a.ell_bits, hk, acc, kl);
#pragma unroll
    for (int kk = 0; kk < K; ++kk) part[((size_t)slot * K + kk) * TP + lp] = acc[kk];
    if (LOSS && slot == 0) kl += fmaxf(a.ell_klc[px], 0.f);   // (negative: the mark of a pixel without counts, no constant)
  };
  auto group_rows = [&](int gi) { return a.ell_off[2 * (tile0 / 64 + gi) + 2] - a.ell_off[2 * (tile0 / 64 + gi)]; };

  int nparts;
  if (pairs) {
    const int gl = wave < 4 ? wave : 7 - wave;           // the longer group of this wave's pair
    const int len_l = group_rows(gl), len_s = group_rows(7 - gl);
    const int half = min(len_l, (len_l + len_s + 1) / 2);   // rows of the longer group its own wave takes
    if (wave < 4) {
      walk_rows(gl, 0, half, 0);
    } else {
      walk_rows(wave, 0, len_s, 0);
      if (half < len_l) walk_rows(gl, half, len_l, 1);
    }
    nparts = 2;
  } else {
    const int nsplit = (NT / 64) / gpw;
    const int gi = wave % gpw, si = wave / gpw;
    const int len = group_rows(gi);
    walk_rows(gi, (int)((long)len * si / nsplit), (int)((long)len * (si + 1) / nsplit), si);
    nparts = nsplit;
  }
  if (cs_late) {   // (uniform) every wave has left the table; the epilogue's barrier orders these stores before its reads
    __syncthreads();
    form_cs();
  }
  h_epilogue<K, true, RULE>(a, part, nparts, TP, tile0, LOSS ? kl : 0.f, cs_lds);
}

// ---- W accumulation ---------------------------------------------------------------------------------
// Workgroup (b, y) = pixel block b (a.pb pixels: ESPM_ELL_PB = 1024 at the full geometry - FULL, a compile-time constant
// there - 128 .. 512 for smaller images) x the channel groups cg = y, y + csplit, ...
// (csplit = gridDim.y; 1 at the headline size, more when the blocks alone do not cover the chip).
// A wave handles 64 channels (one per lane) at a time.  Channel groups are in order of decreasing total
// count: with nw waves, wave w takes the workgroup's groups w, 2 nw - 1 - w, 2 nw + w, ... so the waves carry
// about the same number of entries.  updates.py:38-39, :53, :59.
template <int K, int UNR, bool FULL>
__global__ __launch_bounds__(ESPM_ELL_WTHREADS) void w_accum_ell_kernel(const WAccumArgs a) {
  const int PB = FULL ? ESPM_ELL_PB : a.pb;
  const int PBITS = FULL ? ESPM_ELL_PBITS : a.pbits;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;  // [PB] columns of H
  ell_table_at_lds_zero(tab);
  const int b = blockIdx.x, y = blockIdx.y, csplit = gridDim.y;
  const int nw = (int)blockDim.x >> 6;
  for (int r = threadIdx.x; r < PB; r += (int)blockDim.x) {
    const int q = b * PB + r;
    // (pixels past the end: the row of the last pixel - never referenced by an entry with a count, and positive)
    EllTab<K>::put(tab, PB, r, a.h_t + (size_t)min(q, a.p - 1) * KP);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mine = (a.n_cg - y + csplit - 1) / csplit;  // channel groups of this workgroup
  for (int t = 0; t * nw < mine; ++t) {
    const int i = t * nw + ((t & 1) ? nw - 1 - wave : wave);
    if (i >= mine) continue;
    const int cg = i * csplit + y;
    const int c = a.chan_perm[((size_t)b * a.n_cg + cg) * 64 + lane];
    const float* gsrc = a.gw_s + (size_t)(c < 0 ? 0 : c) * KP;
    float gw[K], acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      gw[kk] = gsrc[kk];
      acc[kk] = 0.f;
    }
    const int32_t* off = a.ell_off + 2 * ((size_t)b * a.n_cg + cg);
    const int beg = off[0], mid = off[1], end = off[2];
    const uint32_t* lrow = a.ell + (size_t)beg * 64 + lane;
    ell_walk<K, UNR>(lrow, mid - beg, EllGetUnit<K>(PB), [&](float, const float (&h)[K]) {
      ell_axpy<K>(acc, h, __builtin_amdgcn_rcpf(ell_dot<K>(h, gw)));
    });
    ell_walk<K, UNR>(lrow + (size_t)(mid - beg) * 64, end - mid, EllGet<K>(tab, PB, PBITS), [&](float x, const float (&h)[K]) {
      const float r = x * __builtin_amdgcn_rcpf(ell_dot<K>(h, gw));
      ell_axpy<K>(acc, h, r);
    });
    if (c >= 0) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) a.a_slab[((size_t)b * K + kk) * a.n_pad + c] = acc[kk];
    }
  }
}

}  // namespace espm
