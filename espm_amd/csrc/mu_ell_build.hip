// Builder of the sparse count store (include/espm_mu.h, x_dtype = ESPM_X_ELL) from the dense pixel-major
// 8-bit store that espm_mu_pack_x writes.  One-time work per fit (a few ms at the headline size), written for
// clarity: every list has ONE owner lane that walks its pixels / channels in order, so no cross-lane
// bookkeeping is needed and the lists come out in ascending index order.
//
//   espm_mu_ell_count : entries and unit elements (count 1) per pixel list and per (pixel block, channel) list, the loss constant
//   espm_mu_ell_plan  : list orders (channels per pixel block, pixels per window: decreasing length), row offsets
//                       (two per group: first unit row, first general row), row totals
//   espm_mu_ell_fill  : the entries.  Unit rows of a group: ESPM_ELL_UNIT_ROWS * floor(min over its 64 lists of the
//                       unit elements / (2 ESPM_ELL_UNIT_ROWS)); 2 * (unit rows) of a list's elements with count 1
//                       go there as index << 4, placed so that the gathers of a wave spread over the LDS banks
//                       (EllBuckets), everything else to the general rows.
#include <mutex>

#include "mu_common.hpp"

namespace espm {

__device__ __forceinline__ int ell_reps(int x, int xmax) { return (x + xmax - 1) / xmax; }

// The non-zero counts of the 256 pixel rows of a workgroup (thread t: row q_mine of the (p, n_pad) 8-bit matrix, -1 for none),
// each in channel order: f(channel, count).  The rows are staged through LDS in chunks of 128 channels: the workgroup fetches
// the chunk of all its rows with consecutive threads on consecutive 8 bytes of a row (rows start and end on 8 bytes: n_pad is a
// multiple of ESPM_NPAD = 8), then every thread walks its own 128 bytes.  Byte by byte and thread by thread, straight from
// memory, the builder spent its time on cache lines it had already evicted (pixel-list count + fill: 13 ms of the build at the
// headline size, 96 ms at C5; now 3.5 and 12.8).
// tile: 256 rows of 33 words (the odd stride keeps the 64 lanes of a wave on distinct banks), s_q: 256 ints.  Every thread
// of the workgroup must call it (barriers).
template <typename F>
__device__ __forceinline__ void ell_block_for_each_count(const uint8_t* __restrict__ x_pm, int n, int n_pad, int q_mine,
                                                         uint32_t* tile, int* s_q, F f) {
  static_assert(ESPM_NPAD % 8 == 0, "8-byte row reads");
  const int t = threadIdx.x;
  s_q[t] = q_mine;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 128) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int id = i * 256 + t, r = id >> 4, c8 = id & 15;
      const int q = s_q[r], c = c0 + 8 * c8;
      uint2 v = make_uint2(0u, 0u);
      if (q >= 0 && c < n_pad) v = *reinterpret_cast<const uint2*>(x_pm + (size_t)q * n_pad + c);
      tile[r * 33 + 2 * c8] = v.x;
      tile[r * 33 + 2 * c8 + 1] = v.y;
    }
    __syncthreads();
    if (q_mine >= 0) {
      for (int w = 0; w < 32; ++w) {
        const uint32_t word = tile[t * 33 + w];
        if (word == 0) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = (int)((word >> (8 * i)) & 255u), c = c0 + 4 * w + i;
          if (x != 0 && c < n) f(c, x);
        }
      }
    }
    __syncthreads();
  }
}

// lane = pixel: walks the pixel's row of the (p, n_pad) 8-bit matrix
// bkt (optional): the list's ones per residue class of the channel index, ESPM_ELL_BUCKETS bytes per pixel - what the fill's placement
// of the unit rows (EllBuckets) otherwise finds out in a pass of its own over X.
__global__ __launch_bounds__(256) void ell_count_h_kernel(const uint8_t* __restrict__ x_pm, int n, int n_pad, int p, int p_pad,
                                                          int xmax, int unit_ok, int32_t* __restrict__ cnt_px,
                                                          float* __restrict__ klc, uint8_t* __restrict__ bkt) {
  __shared__ uint32_t s_tile[256 * 33];
  __shared__ int s_q[256];
  __shared__ uint8_t s_bk[ESPM_ELL_BUCKETS * 256];
  const int q = blockIdx.x * 256 + threadIdx.x;   // (p_pad is a multiple of 256: every thread has a slot)
  int cnt = 0, ones = 0;
  double corr = 0.0;
  const double lxm = (double)xmax * log2((double)xmax);
  if (bkt)
    for (int i = 0; i < ESPM_ELL_BUCKETS; ++i) s_bk[i * 256 + threadIdx.x] = 0;
  ell_block_for_each_count(x_pm, n, n_pad, q < p ? q : -1, s_tile, s_q, [&](int c, int x) {
    const int r = ell_reps(x, xmax);
    cnt += r;
    ones += x == 1;
    if (bkt && x == 1) s_bk[(c & (ESPM_ELL_BUCKETS - 1)) * 256 + threadIdx.x] += 1;   // (8 bits: as EllBuckets' counters)
    if (r > 1) {  // x log2 x - sum over its entries of x_i log2 x_i
      const int rest = x - (r - 1) * xmax;
      corr += (double)x * log2((double)x) - (double)(r - 1) * lxm - (double)rest * log2((double)(rest > 1 ? rest : 1));
    }
  });
  if (q >= p_pad) return;
  cnt_px[q] = cnt;
  cnt_px[p_pad + q] = unit_ok ? ones : 0;
  klc[q] = (float)corr;
  if (bkt)
    for (int i = 0; i < ESPM_ELL_BUCKETS; ++i) bkt[(size_t)q * ESPM_ELL_BUCKETS + i] = s_bk[i * 256 + threadIdx.x];
}

// lane = channel: walks the pixels of one block
// (bkt: as ell_count_h_kernel's, per (block, channel) list and residue class of the pixel's index inside the block)
__global__ __launch_bounds__(256) void ell_count_w_kernel(const uint8_t* __restrict__ x_pm, int n, int n_pad, int p, int ncol,
                                                          int xmax, int pb, int32_t* __restrict__ cnt_bc, uint8_t* __restrict__ bkt) {
  __shared__ uint8_t s_bk[ESPM_ELL_BUCKETS * 256];
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= ncol) return;   // (no barrier below)
  int cnt = 0, ones = 0;
  if (bkt)
    for (int i = 0; i < ESPM_ELL_BUCKETS; ++i) s_bk[i * 256 + threadIdx.x] = 0;
  if (c < n) {
    const int q0 = b * pb, q1 = min(p, q0 + pb);
    for (int q = q0; q < q1; ++q) {
      const int x = x_pm[(size_t)q * n_pad + c];
      if (x) cnt += ell_reps(x, xmax);
      ones += x == 1;
      if (bkt && x == 1) s_bk[((q - q0) & (ESPM_ELL_BUCKETS - 1)) * 256 + threadIdx.x] += 1;
    }
  }
  cnt_bc[(size_t)b * ncol + c] = cnt;
  cnt_bc[((size_t)gridDim.x + b) * ncol + c] = ones;
  if (bkt)
    for (int i = 0; i < ESPM_ELL_BUCKETS; ++i) bkt[((size_t)b * ncol + c) * ESPM_ELL_BUCKETS + i] = s_bk[i * 256 + threadIdx.x];
}

// Row offsets of `count` list groups, one workgroup of 1024 threads: f(i, unit) returns the rows of group i and sets
// the unit rows among them; out[2 i] = first (unit) row, out[2 i + 1] = first general row, out[2 count] = all rows.
template <typename F>
__device__ void block_scan_rows(int count, int32_t* out, long long* total, long long* s_sums, F f) {
  const int t = threadIdx.x, nt = blockDim.x;
  const int chunk = (count + nt - 1) / nt;
  const int i0 = min(count, t * chunk), i1 = min(count, i0 + chunk);
  auto clamp31 = [](long long v) { return (int32_t)(v < 0x7fffffffLL ? v : 0x7fffffffLL); };
  long long s = 0;
  int unit = 0;
  for (int i = i0; i < i1; ++i) s += f(i, unit);
  s_sums[t] = s;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int i = 0; i < nt; ++i) {
      const long long v = s_sums[i];
      s_sums[i] = run;
      run += v;
    }
    *total = run;
    out[2 * count] = clamp31(run);
  }
  __syncthreads();
  long long run = s_sums[t];
  for (int i = i0; i < i1; ++i) {
    const long long rows = f(i, unit);
    out[2 * i] = clamp31(run);
    out[2 * i + 1] = clamp31(run + unit);
    run += rows;
  }
  __syncthreads();
}

// Orders by decreasing list length (stable: ties keep the lower index first - what torch.argsort(descending=True,
// stable=True) gives).  Workgroups [0, nblk): the channels of pixel block b -> chan_perm[b]; workgroups
// [nblk, nblk + nwin): the pixels of window w -> pix_perm.
__global__ __launch_bounds__(1024) void ell_order_kernel(const int32_t* __restrict__ cnt_px, const int32_t* __restrict__ cnt_bc, int n,
                                                         int n_cg, int nblk, int win, int32_t* __restrict__ chan_perm,
                                                         int32_t* __restrict__ pix_perm) {
  extern __shared__ int32_t s_cnt[];  // [max(n, win)]
  const int ncol = n_cg * 64;
  const bool chan = (int)blockIdx.x < nblk;
  const int count = chan ? n : win;
  const int32_t* src = chan ? cnt_bc + (size_t)blockIdx.x * ncol : cnt_px + (size_t)(blockIdx.x - nblk) * win;
  int32_t* dst = chan ? chan_perm + (size_t)blockIdx.x * ncol : pix_perm + (size_t)(blockIdx.x - nblk) * win;
  for (int i = threadIdx.x; i < count; i += blockDim.x) s_cnt[i] = src[i];
  if (chan)
    for (int s = n + threadIdx.x; s < ncol; s += blockDim.x) dst[s] = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const int mine = s_cnt[i];
    int rank = 0;
    for (int o = 0; o < count; ++o) {
      const int v = s_cnt[o];
      rank += (v > mine) || (v == mine && o < i);
    }
    dst[rank] = i;
  }
}

// rows of a group whose longest list has m entries and whose poorest list has u unit elements
__device__ __forceinline__ long long ell_group_rows(int m, int u, int& unit) {
  unit = u / (2 * ESPM_ELL_UNIT_ROWS) * ESPM_ELL_UNIT_ROWS;
  return (long long)unit + (m - 2 * unit + 1) / 2;
}

// (longest list, fewest unit elements) of every list group, one wave per group: H groups [0, ngrp), then the W groups
// (block, channel group); left in the offset arrays themselves (h_off[2 g], h_off[2 g + 1]), which ell_offsets_kernel then
// turns into offsets in place.  (Inside the scan, with one thread per few groups and 64 dependent loads each, this took
// 3 ms of the headline's build and 16 ms of C5's.)
__global__ __launch_bounds__(256) void ell_group_extent_kernel(const int32_t* __restrict__ cnt_px, const int32_t* __restrict__ cnt_bc,
                                                               int n_cg, int nblk, int ngrp, int win,
                                                               const int32_t* __restrict__ chan_perm, const int32_t* __restrict__ pix_perm,
                                                               int32_t* __restrict__ h_off, int32_t* __restrict__ w_off) {
  const int lane = threadIdx.x & 63;
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ncol = n_cg * 64, p_pad = ngrp * 64;
  if (g >= (long long)ngrp + (long long)nblk * n_cg) return;   // (whole waves)
  int m, u;
  int32_t* out;
  if (g < ngrp) {
    const int w0 = ((int)g * 64) / win * win;
    const int q = w0 + pix_perm[g * 64 + lane];
    m = cnt_px[q];
    u = cnt_px[p_pad + q];
    out = h_off + 2 * g;
  } else {
    const long long i = g - ngrp;
    const int b = (int)(i / n_cg);
    const int c = chan_perm[i * 64 + lane];
    m = c >= 0 ? cnt_bc[(size_t)b * ncol + c] : 0;
    u = c >= 0 ? cnt_bc[((size_t)nblk + b) * ncol + c] : 0;
    out = w_off + 2 * i;
  }
  for (int o = 32; o; o >>= 1) {
    m = max(m, __shfl_xor(m, o));
    u = min(u, __shfl_xor(u, o));
  }
  if (lane == 0) {
    out[0] = m;
    out[1] = u;
  }
}

__global__ __launch_bounds__(1024) void ell_offsets_kernel(int n_cg, int nblk, int ngrp, int32_t* __restrict__ h_off,
                                                           int32_t* __restrict__ w_off, long long* __restrict__ rows) {
  __shared__ long long s_sums[1024];
  // H lists of slot group g, then the W lists of (block, channel group): extents from ell_group_extent_kernel, replaced by
  // the offsets (a thread reads an entry before it overwrites it)
  block_scan_rows(ngrp, h_off, &rows[0], s_sums, [&](int g, int& unit) { return ell_group_rows(h_off[2 * g], h_off[2 * g + 1], unit); });
  block_scan_rows(nblk * n_cg, w_off, &rows[1], s_sums, [&](int i, int& unit) { return ell_group_rows(w_off[2 * i], w_off[2 * i + 1], unit); });
}

// Placement of a list's ones in its unit rows so that the table gathers of a wave spread over the LDS banks.
// The kernels gather a table row with ds_read_b128 (components 0..3, 16-byte stride) and, from 5 components on, its further
// components with ds_read_b32 / b64 from a second array (4- or 8-byte stride).  MI355X_MICROARCH.md, LDS: the 16-byte read is served
// in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: every group holds each lane number
// mod 16 once - and two lanes of a group collide when their rows share index mod 16 (the 16-byte slot of the 256-byte bank row);
// the 4-byte read is served in the two halves of 32 lanes, banks = index mod 32.  Lane l therefore puts a one of bucket
// q = index mod ESPM_ELL_BUCKETS at a position p with (p + l) mod ESPM_ELL_BUCKETS = q - the r-th one of the bucket (ascending
// index) at p = ((q - l) mod B) + B r while that is inside the unit entries.  With B = 32 the 32 lanes of a half read 32
// different banks AND the 16 lanes of a 16-byte group 16 different slots; with B = 16 lanes l and l + 16 hold rows of
// the same index mod 16, which collide mod 32 every second time: the 4-byte read of nearly every unit entry takes two passes
// (5.6 M of the fused kernel's 19.8 M conflict cycles per launch by the counts of profiles/r04r_pmc_sq2*).
// Positions a short bucket leaves empty ("holes") are filled, in order of (bucket, position), by the ones that did not fit
// their bucket, in order of index; the rest joins the general rows.  Every lane decides alone, from the bucket counts of its own
// list (at most 255 ones per bucket: unit rows exist for n <= ESPM_ELL_UNIT_MAX_N = 4080 channels / 1024 pixels per block - 8-bit counters;
// with 4096 a full residue class of 256 ones wrapped its counter to 0 and the overflow ones were placed on occupied positions).
// MEASURED (round 4, profiles/r04u_buckets_ab_*.log): B = 32 buys nothing - headline 137.1 (16) against 137.5 us (32), k = 3 113.2 / 113.8, a
// 64-row shard 31.7 / 31.8, configuration 5's 128-row shard 110.3 / 111.5: the second pass of the 4-byte reads was not on the critical
// path, and half-size buckets leave more holes for the 16-byte reads.  16 stays the default.
// (ESPM_ELL_BUCKETS: include/espm_mu.h - the count step can hand the histograms to the fill, espm_mu_ell_count_hist)
struct EllBuckets {
  static constexpr int B = ESPM_ELL_BUCKETS;
  static_assert(B == 16 || B == 32, "16 or 32 buckets");
  static_assert((ESPM_ELL_UNIT_MAX_N + B - 1) / B <= 255 && (ESPM_ELL_PB + B - 1) / B <= 255, "8-bit bucket counters: at most 255 ones of a list in one residue class");
  uint8_t* cnt;     // [B][nthreads]: ones per bucket (pass 1)
  uint8_t* seen;    // [B][nthreads]: ones of the bucket met so far (pass 2)
  int nt, t;
  static constexpr size_t bytes(int nthreads) { return (size_t)2 * B * nthreads; }
  __device__ __forceinline__ uint8_t& c(int q) { return cnt[q * nt + t]; }
  __device__ __forceinline__ uint8_t& s(int q) { return seen[q * nt + t]; }
  __device__ __forceinline__ void clear() {
    for (int q = 0; q < B; ++q) { c(q) = 0; s(q) = 0; }
  }
  __device__ __forceinline__ void count(int idx) { c(idx & (B - 1)) += 1; }
  // positions of bucket q in a lane's `units` unit entries: base, base + B, ...
  static __device__ __forceinline__ int base(int q, int lane) { return (q - lane) & (B - 1); }
  static __device__ __forceinline__ int nslots(int q, int lane, int units) {
    const int b0 = base(q, lane);
    return b0 < units ? (units - b0 + B - 1) / B : 0;
  }
  // after pass 1: the holes of all buckets
  __device__ __forceinline__ int plan(int lane, int units) {
    int run = 0;
    for (int q = 0; q < B; ++q) run += max(0, nslots(q, lane, units) - (int)c(q));
    return run;
  }
  // pass 2: unit position of this one, or -1 when it belongs to the general rows
  __device__ __forceinline__ int place(int idx, int lane, int units, int holes, int& overflow) {
    const int q = idx & (B - 1);
    const int r = s(q);
    s(q) = (uint8_t)(r + 1);
    if (r < nslots(q, lane, units)) return base(q, lane) + B * r;
    int k = overflow++;
    if (k >= holes) return -1;
    for (int qh = 0; qh < B; ++qh) {   // hole k in order of (bucket, position): the prefix sums are re-formed here - one in ten ones overflows
      const int hq = max(0, nslots(qh, lane, units) - (int)c(qh));
      if (k < hq) return base(qh, lane) + B * ((int)c(qh) + k);
      k -= hq;
    }
    return -1;
  }
};

__device__ __forceinline__ void ell_put(uint16_t* base16, size_t row0, int j, int lane, uint32_t entry) {
#ifdef ESPM_EXPERIMENT_FILL_NO_STORE   // TIMING ONLY (tools/analysis): what the fill's 2-byte stores cost - one entry per lane into a line of its own
  if (entry == 0xffffffffu) base16[0] = 0;
  return;
#endif
  base16[((row0 + (size_t)(j >> 1)) * 64 + lane) * 2 + (j & 1)] = (uint16_t)entry;
}

// lane = list slot (a wave = one group of 64 slots); slot -> pixel through pix_perm
__global__ __launch_bounds__(256) void ell_fill_h_kernel(const uint8_t* __restrict__ x_pm, int n, int n_pad, int p, int p_pad,
                                                         int cbits, int win, const int32_t* __restrict__ pix_perm,
                                                         const int32_t* __restrict__ h_off, uint32_t* __restrict__ ell_h,
                                                         const uint8_t* __restrict__ bkt) {
  __shared__ uint8_t s_b[EllBuckets::bytes(256)];
  __shared__ uint32_t s_tile[256 * 33];
  __shared__ int s_q[256];
  const int slot = blockIdx.x * 256 + threadIdx.x;
  int q = -1;   // (a slot beyond p_pad, a pixel beyond p: no list, but the thread still helps to stage the rows)
  if (slot < p_pad) {
    q = slot / win * win + pix_perm[slot];
    if (q >= p) q = -1;
  }
  const int xmax = (1 << (16 - cbits)) - 1;
  uint16_t* base16 = reinterpret_cast<uint16_t*>(ell_h);
  size_t row0 = 0, row1 = 0;
  if (q >= 0) {
    row0 = (size_t)h_off[2 * (slot >> 6)];
    row1 = (size_t)h_off[2 * (slot >> 6) + 1];
  }
  const int units = 2 * (int)(row1 - row0);
  const int lane = slot & 63;
  EllBuckets b{s_b, s_b + EllBuckets::B * 256, 256, (int)threadIdx.x};
  int holes = 0;
  b.clear();
  if (bkt) {   // (uniform) the count step left the ones per residue class: no first pass over X
    if (q >= 0 && units)
      for (int i = 0; i < EllBuckets::B; ++i) b.c(i) = bkt[(size_t)q * EllBuckets::B + i];
  } else {
    ell_block_for_each_count(x_pm, n, n_pad, q, s_tile, s_q, [&](int c, int x) {
      if (x == 1 && units) b.count(c);
    });
  }
  if (units) holes = b.plan(lane, units);
  int j = 0, overflow = 0;
  ell_block_for_each_count(x_pm, n, n_pad, q, s_tile, s_q, [&](int c, int x) {
    if (x == 1 && units) {
      const int pos = b.place(c, lane, units, holes, overflow);
      if (pos >= 0) {
        ell_put(base16, row0, pos, lane, (uint32_t)c << 4);
        return;
      }
    }
    while (x > 0) {
      const int v = x > xmax ? xmax : x;
      ell_put(base16, row1, j++, lane, ((uint32_t)v << cbits) | (uint32_t)c);
      x -= v;
    }
  });
}

// lane = channel slot (a wave = one channel group of one pixel block)
// (ESPM_ELL_FILLW_WAVES channel groups of one block per workgroup: the waves read the same pixel rows at about the same
//  time, and a lane's byte drags in the cache line around it - with a group per workgroup every row was fetched 32 times)
#ifndef ESPM_ELL_FILLW_WAVES
#define ESPM_ELL_FILLW_WAVES 4
#endif
// x_cm (optional): the same counts channel-major inside tiles of ESPM_PPAD pixels, [tile][channel row, n_cm of them][ESPM_PPAD]
// (espm_mu_pack_x's other output), zero beyond p: a lane then fetches 16 pixels of ITS channel with one 16-byte load instead of
// 16 byte loads that each drag a cache line of the pixel's row through the memory pipe (3.7 -> 1.x ms at the headline size: this
// kernel was the longest of the build).
__global__ __launch_bounds__(64 * ESPM_ELL_FILLW_WAVES) void ell_fill_w_kernel(const uint8_t* __restrict__ x_pm, const uint8_t* __restrict__ x_cm, int n_cm,
                                                        int n_pad, int p, int n_cg, int pb, int pbits,
                                                        const int32_t* __restrict__ chan_perm, const int32_t* __restrict__ w_off,
                                                        uint32_t* __restrict__ ell_w, const uint8_t* __restrict__ bkt) {
  constexpr int NT = 64 * ESPM_ELL_FILLW_WAVES;
  __shared__ uint8_t s_b[EllBuckets::bytes(NT)];
  const int b = blockIdx.x, cg = blockIdx.y * ESPM_ELL_FILLW_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cg >= n_cg) return;   // (whole waves; no barrier below)
  const int c = chan_perm[((size_t)b * n_cg + cg) * 64 + lane];
  if (c < 0) return;
  const int xmax = (1 << (16 - pbits)) - 1;
  uint16_t* base16 = reinterpret_cast<uint16_t*>(ell_w);
  const size_t row0 = (size_t)w_off[2 * ((size_t)b * n_cg + cg)], row1 = (size_t)w_off[2 * ((size_t)b * n_cg + cg) + 1];
  const int units = 2 * (int)(row1 - row0);
  const int q0 = b * pb, q1 = min(p, q0 + pb);
  EllBuckets bk{s_b, s_b + EllBuckets::B * NT, NT, (int)threadIdx.x};
  int holes = 0;
  // the channel's counts over the block's pixels, 16 loads in flight at a time (one by one, each waited for, this kernel
  // was the longest of the build: the lanes' bytes of a pixel lie all over its row)
  auto for_each_count = [&](auto f) {
    if (x_cm) {   // (uniform) 64 pixels = four 16-byte loads in flight; blocks start on multiples of 128 pixels, tiles hold ESPM_PPAD
      for (int qb = q0; qb < q1; qb += 64) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = qb + 16 * u;
          v[u] = q < q1 ? *reinterpret_cast<const uint4*>(x_cm + ((size_t)(q / ESPM_PPAD) * n_cm + c) * ESPM_PPAD + (q % ESPM_PPAD)) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if (w[d] == 0u) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int x = (int)((w[d] >> (8 * e)) & 255u);
              if (x != 0) f(qb + 16 * u + 4 * d + e - q0, x);
            }
          }
        }
      }
      return;
    }
    for (int qb = q0; qb < q1; qb += 16) {
      int x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = qb + u < q1 ? (int)x_pm[(size_t)(qb + u) * n_pad + c] : 0;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (x[u] != 0) f(qb + u - q0, x[u]);
    }
  };
  if (units) {
    bk.clear();
    if (bkt) {   // (uniform) from the count step
      for (int i = 0; i < EllBuckets::B; ++i) bk.c(i) = bkt[((size_t)b * n_cg * 64 + c) * EllBuckets::B + i];
    } else {
      for_each_count([&](int i, int x) {
        if (x == 1) bk.count(i);
      });
    }
    holes = bk.plan(lane, units);
  }
  int j = 0, overflow = 0;
  for_each_count([&](int i, int x) {
    if (x == 1 && units) {
      const int pos = bk.place(i, lane, units, holes, overflow);
      if (pos >= 0) {
        ell_put(base16, row0, pos, lane, (uint32_t)i << 4);
        return;
      }
    }
    while (x > 0) {
      const int v = x > xmax ? xmax : x;
      ell_put(base16, row1, j++, lane, ((uint32_t)v << pbits) | (uint32_t)i);
      x -= v;
    }
  });
}

// The pixel lists' and the channel lists' kernels of a step do not depend on each other and neither fills the device's memory
// pipes (one owner lane per list, byte-granular reads): the channel lists' kernel runs on a side stream of the device, forked
// from and joined to the caller's stream by events.  ESPM_ELL_BUILD_SIDE=0: one after the other on the caller's stream (A/B).
struct SideStream {
  hipStream_t stream = nullptr;   // the device's side stream (shared by the host threads that build on this device: work on it is ordered)
  hipEvent_t fork = nullptr, join = nullptr;   // this call's own events (two threads building at once must not share them)
};
static hipStream_t side_stream_of_device() {
  static const bool enabled = [] { const char* e = getenv("ESPM_ELL_BUILD_SIDE"); return !(e && e[0] == '0'); }();
  if (!enabled) return nullptr;
  static std::mutex mu;
  static hipStream_t per_device[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!per_device[dev] && hipStreamCreateWithFlags(&per_device[dev], hipStreamNonBlocking) != hipSuccess) per_device[dev] = nullptr;
  return per_device[dev];
}
// the stream the second kernel of a pair goes to: the side stream, made to wait for what the caller's stream holds so far - or the caller's own
static hipStream_t side_fork(SideStream& s, hipStream_t stream) {
  s.stream = side_stream_of_device();
  if (!s.stream) return stream;
  if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess ||
      hipEventRecord(s.fork, stream) != hipSuccess || hipStreamWaitEvent(s.stream, s.fork, 0) != hipSuccess) {
    (void)hipGetLastError();   // (the pair then runs on the caller's stream: the failed call's error is not the launch's)
    s.stream = nullptr;
    return stream;
  }
  return s.stream;
}
static int side_join(SideStream& s, hipStream_t stream) {
  int rc = ESPM_OK;
  if (s.stream) {
    rc = check_hip(hipEventRecord(s.join, s.stream), "ell build: side stream");
    if (!rc) rc = check_hip(hipStreamWaitEvent(stream, s.join, 0), "ell build: side stream");
  }
  if (s.fork) (void)hipEventDestroy(s.fork);   // (released once the work that refers to them has completed)
  if (s.join) (void)hipEventDestroy(s.join);
  return rc;
}

int launch_ell_count(const uint8_t* x_pm, int n, int n_pad, int p, int p_pad, int cbits, int n_cg, int nblk, int pb,
                     int32_t* cnt_px, int32_t* cnt_bc, float* klc, hipStream_t stream, uint8_t* bkt_px, uint8_t* bkt_bc) {
  const int xmax_h = (1 << (16 - cbits)) - 1, xmax_w = (1 << (16 - ell_pbits(pb))) - 1;
  // a unit entry holds index << 4 in 16 bits
  SideStream side;
  const hipStream_t s2 = side_fork(side, stream);   // (before the first kernel: the side stream waits for what precedes the pair only)
  hipLaunchKernelGGL(ell_count_h_kernel, dim3((p_pad + 255) / 256), dim3(256), 0, stream, x_pm, n, n_pad, p, p_pad, xmax_h,
                     n <= ESPM_ELL_UNIT_MAX_N ? 1 : 0, cnt_px, klc, bkt_px);
  hipLaunchKernelGGL(ell_count_w_kernel, dim3(nblk, (n_cg * 64 + 255) / 256), dim3(256), 0, s2, x_pm, n, n_pad, p,
                     n_cg * 64, xmax_w, pb, cnt_bc, bkt_bc);
  const int rc = check_hip(hipGetLastError(), "ell_count launch");
  const int rc2 = side_join(side, stream);
  return rc ? rc : rc2;
}

int launch_ell_plan(const int32_t* cnt_px, const int32_t* cnt_bc, int n, int n_cg, int nblk, int p_pad, int win,
                    int32_t* chan_perm, int32_t* pix_perm, int32_t* h_off, int32_t* w_off, long long* rows, hipStream_t stream) {
  const size_t lds = (size_t)(n > win ? n : win) * sizeof(int32_t);
  if (lds > 64 * 1024) {
    if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(ell_order_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "ell_plan"))
      return rc;
  }
  hipLaunchKernelGGL(ell_order_kernel, dim3(nblk + p_pad / win), dim3(1024), lds, stream, cnt_px, cnt_bc, n, n_cg, nblk, win,
                     chan_perm, pix_perm);
  const long long groups = (long long)(p_pad / 64) + (long long)nblk * n_cg;
  hipLaunchKernelGGL(ell_group_extent_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, stream, cnt_px, cnt_bc, n_cg, nblk,
                     p_pad / 64, win, chan_perm, pix_perm, h_off, w_off);
  hipLaunchKernelGGL(ell_offsets_kernel, dim3(1), dim3(1024), 0, stream, n_cg, nblk, p_pad / 64, h_off, w_off, rows);
  return check_hip(hipGetLastError(), "ell_plan launch");
}

int launch_ell_fill(const uint8_t* x_pm, int n, int n_pad, int p, int p_pad, int cbits, int n_cg, int nblk, int win, int pb,
                    const int32_t* chan_perm, const int32_t* pix_perm, const int32_t* h_off, const int32_t* w_off,
                    uint32_t* ell_h, uint32_t* ell_w, hipStream_t stream, const uint8_t* x_cm, int n_cm, const uint8_t* bkt_px,
                    const uint8_t* bkt_bc) {
  SideStream side;
  const hipStream_t s2 = side_fork(side, stream);
  hipLaunchKernelGGL(ell_fill_h_kernel, dim3((p_pad + 255) / 256), dim3(256), 0, stream, x_pm, n, n_pad, p, p_pad, cbits, win,
                     pix_perm, h_off, ell_h, bkt_px);
  hipLaunchKernelGGL(ell_fill_w_kernel, dim3(nblk, (n_cg + ESPM_ELL_FILLW_WAVES - 1) / ESPM_ELL_FILLW_WAVES), dim3(64 * ESPM_ELL_FILLW_WAVES), 0, s2, x_pm, x_cm, n_cm, n_pad, p, n_cg, pb, ell_pbits(pb), chan_perm, w_off, ell_w, bkt_bc);
  const int rc = check_hip(hipGetLastError(), "ell_fill launch");
  const int rc2 = side_join(side, stream);
  return rc ? rc : rc2;
}

}  // namespace espm
