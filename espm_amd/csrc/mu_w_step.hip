// W update of the SmoothNMF multiplicative rule.
//
// espm/estimators/updates.py:38-76 computes  W' = max( W * (G^T (X / (GWH)) H^T) / (colsum(G) rowsum(H)^T + nu), eps ).
// Here the (n, p) ratio R = X / (GW H) is never stored and the association is G^T (R H^T):
//   w_accum : A_b = sum_{j in pixel block b} R[:, j] H[:, j]^T      lanes <-> channels, X pixel-major,
//             one 16-byte coalesced load per lane and pixel, H[:, j] wave-uniform (scalar cache),
//             the contraction over pixels accumulates in registers - no cross-lane traffic.
//   w_reduce: A = sum_b A_b in fixed order (bit-reproducible, unlike float atomics).
//   w_finish: numerator / denominator, optional simplex over the columns of W with the reference's
//             global-stop bisection (dicotomy.py:111-173), clamp, fixed_W, then GW = G W for the
//             next half step, its column sums, and rel_W (base.py:323).
#include <atomic>
#include <stdio.h>
#include <stdlib.h>

#include "mu_w_kernel.hpp"
#include "mu_w_mfma_kernel.hpp"
#include "mu_xchg.hpp"

#ifndef ESPM_MFMA_MIN_K
#define ESPM_MFMA_MIN_K 9   // measured at the headline image, 8-bit store: k = 7 493 vs 499 us per iteration, k = 8 519 vs 522 (no gain), k = 12 642 vs 761, k = 16 750 vs 992
#endif

namespace espm {

// Slab reduction A = sum_b A_b in ONE pass and in a fixed order (bit-reproducible; no float atomics):
// a workgroup owns 32 consecutive entries of A; its 256 threads are 32 entries x 8 slab groups, group g
// sums the slabs b = g, g + 8, ... with 8 independent partial sums (loads in flight), then the 8 groups are
// combined through LDS in group order.  One extra workgroup (when `fin` is set) reduces the H-step's
// per-workgroup records at the same time (h_finalize_body), which saves a dependent launch per iteration.
struct WReduceArgs {
  const float* slab;
  float* out;
  int nblk, total, nred_blocks, fuse_finalize;
  HFinalizeArgs fin;
  // sharded image: `out` is the A block of this rank's record and the extra workgroup also copies the first and
  // the last owned image row of the new H behind the statistics (shard_pack_kernel's job, without its launch)
  const float* halo_h;
  float* halo_top;
  float* halo_bot;
  int halo_k, halo_nx, halo_ny, halo_ppad;
  // simplex over W with G = identity (w_simplex_update_kernel): the workgroup also leaves, for its 32 entries of component
  // kk = entry / n_pad (n_pad a multiple of 32), what the bracket and the root of that component's multiplier need -
  // sum, maximum and count of the positive numerators W A (dicotomy.py:29-49) - in bparts[3 * workgroup ..]; else null
  const float* bw_old;
  double* bparts;
  int bn, bk, bn_pad;
};

__global__ __launch_bounds__(256) void w_reduce_kernel(const WReduceArgs a) {
  __shared__ double fscratch[5 * (ESPM_HP_NSCALAR + 2 * KP + 1)];
  __shared__ float s_part[8][32];
  if ((int)blockIdx.x >= a.nred_blocks) {  // the extra workgroups: one value of the record reduction each (h_finalize_one); the boundary rows with the one that needs no records
    const int job = (int)blockIdx.x - a.nred_blocks;
    if (a.halo_top && job == H_FINALIZE_NV) {
      for (int e = threadIdx.x; e < a.halo_k * a.halo_ny; e += 256) {
        const int kk = e / a.halo_ny, j = e - kk * a.halo_ny;
        a.halo_top[e] = a.halo_h[(size_t)kk * a.halo_ppad + j];
        a.halo_bot[e] = a.halo_h[(size_t)kk * a.halo_ppad + (size_t)(a.halo_nx - 1) * a.halo_ny + j];
      }
    }
    h_finalize_one(a.fin, job, fscratch);
    return;
  }
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + col;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (e < a.total) {
    int b = grp;
    for (; b + 56 < a.nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += a.slab[(size_t)(b + 8 * u) * a.total + e];
    }
    for (int u = 0; b < a.nblk; b += 8, ++u) acc[u & 7] += a.slab[(size_t)b * a.total + e];
  }
  s_part[grp][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (grp == 0) {   // (32 lanes: half of wave 0)
    float t = 0.f;
    if (e < a.total) {
#pragma unroll
      for (int g = 0; g < 8; ++g) t += s_part[g][col];
      a.out[e] = t;
    }
    if (a.bparts) {
      const int kk = e / a.bn_pad, c = e - kk * a.bn_pad;
      const float num = (e < a.total && c < a.bn) ? a.bw_old[(size_t)c * a.bk + kk] * t : 0.f;   // updates.py:59 (G = identity)
      double sum = num > 0.f ? (double)num : 0.0, cnt = num > 0.f ? 1.0 : 0.0;
      float mx = fmaxf(num, 0.f);
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {   // (fixed order: the same partials run to run)
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      }
      if (col == 0) {
        a.bparts[3 * (size_t)blockIdx.x] = sum;
        a.bparts[3 * (size_t)blockIdx.x + 1] = (double)mx;
        a.bparts[3 * (size_t)blockIdx.x + 2] = cnt;
      }
    }
  }
}

constexpr int WF_THREADS = 1024;
#ifndef ESPM_WF_FEW_THREADS
#define ESPM_WF_FEW_THREADS 512
#endif
constexpr int WF_FEW_THREADS = ESPM_WF_FEW_THREADS;   // threads of the register-resident W finish for G = identity, k <= WF_HALF_MAX_K (A/B: 256)
#ifndef ESPM_WF_HALF_MAX_K
#define ESPM_WF_HALF_MAX_K 8
#endif
constexpr int WF_HALF_MAX_K = ESPM_WF_HALF_MAX_K;   // component counts up to which the register-resident W finish also exists with 512 threads

// ---- Slab (or rank-record) reduction with the W update folded in: G = identity, no simplex_W -------------------
// When W' needs nothing global beyond the row sums of the new H (updates.py:58-60, :70-76 with G = I and no
// simplex over W), the update of an entry of W only needs the matching entry of A = sum of the sources: the
// workgroups that reduce the sources finish "their" entries of W right away - W' = max(W A / rowsum(H'), eps),
// fixed_W, the row of G W' for the next half step - instead of one workgroup doing all of W afterwards.  What IS
// global (column sums of G W', mean of W' for rel_W) leaves as per-workgroup partials for w_update_tail_kernel.
//   sources: `nsrc` arrays of (k, n_pad) floats, `src_stride` bytes apart: the slabs of the W accumulation, or
//            the A blocks of the ranks' records (sharded image), always summed in the same fixed order.
//   row sums of the new H: from the H-step's per-workgroup records (same order of operations as h_finalize_body,
//            so the value equals hstat's), or the sum over the ranks' records (which this kernel also turns into
//            the global hstat, like shard_combine).
// Workgroup (kk, j) owns channels [32 j, 32 j + 32) of component kk; one extra workgroup runs h_finalize_body.
struct WUpdateArgs {
  const unsigned char* src;
  size_t src_stride;
  int nsrc, n, n_pad, k, nbk;
  float* a_out;
  const double* hpart;          // slab mode: H-step records (field-major), nblk_h of them
  const double* hstat_rs;       // slab mode without a riding finalize: the statistics of the new H are already reduced
  int nblk_h;
  size_t rec_hstat_off;         // records mode (hpart == null): byte offset of the 16 statistics inside a record
  double* hstat_out;            // records mode: global row sums / maxima of the new H
  const float* w_old;
  float* w_new;
  const float* fixed_w;
  const float* breg_sr;         // Bregman variant (updates.py:40-48): per-channel sums of the stored X, else null
  float pg_gamma_w;             // > 0: projected-gradient step (updates.py:353-370)
  int pg_track;                 // its linesearch term goes to parts[2 nwg + workgroup]
  float* gw_s;
  double* parts;                // [2][k * nbk]: partial column sum of G W' (component of the workgroup), partial sum of W'
  float log_shift, gw_floor, xscale;
  int fuse_finalize;
  HFinalizeArgs fin;
};

// The update of the 32 entries of W (component kk, channels c of the lanes that own one) from their summed A and the row
// sum rs of the new H, by wave 0 of a reduction workgroup; the per-workgroup partials of what is global go to a.parts.
__device__ __forceinline__ void w_update_preload(const WUpdateArgs& a, int kk, int c, bool mine, float& wo, float& fx) {
  wo = 1.f;
  fx = -1.f;
  if (mine && c < a.n) {
    wo = a.w_old[(size_t)c * a.k + kk];
    if (a.fixed_w) fx = a.fixed_w[(size_t)c * a.k + kk];
  }
}
// wo_pre / fx_pre: the entry's old value and fixed value (negative: none), requested by the caller at its start - here they would
// be one more trip to memory at the end of a kernel that is nothing but latency.
__device__ __forceinline__ void w_update_entries(const WUpdateArgs& a, int kk, int c, int e, bool owns, float t, double rs, int nwg, int wg,   // wg: index of the reduction workgroup (its slot of the partials)
                                                 float wo_pre, float fx_pre) {
  double cs = 0.0, sw = 0.0, qw = 0.0;
  if (owns) {
    a.a_out[e] = t;
    if (c < a.n) {
      const float wo = wo_pre;
      float v;
      if (a.pg_gamma_w > 0.f) {  // W - grad / gamma with grad = rowsum(H) - (X / GWH) H^T (G = I), updates.py:353-362
        v = fmaxf(wo - ((float)rs - t) / a.pg_gamma_w, a.log_shift);
        const double dw = (double)v - (double)wo;   // (fixed_W is not part of a projected-gradient fit, smooth_nmf.py:430-437)
        qw = dw * (double)((float)rs - t) + (double)a.pg_gamma_w * dw * dw;
      } else if (a.breg_sr) {  // W' = sR W / ((rowsum(H) - (X / GWH) H^T) W + sR), updates.py:41-48
        const float sr = a.xscale * a.breg_sr[c];
        v = fmaxf((sr * wo) / (((float)rs - t) * wo + sr), a.log_shift);
      } else {
        v = fmaxf((wo * t) / (float)rs, a.log_shift);   // updates.py:59-60, :70-72 (G = I: colsum(G) = 1)
      }
      if (fx_pre >= 0.f) v = fx_pre;                        // updates.py:75-76
      a.w_new[(size_t)c * a.k + kk] = v;
      const float gv = fmaxf(v, a.gw_floor);
      a.gw_s[(size_t)c * KP + kk] = gv * (1.f / a.xscale);
      cs = (double)gv;
      sw = (double)v;
    } else {
      a.gw_s[(size_t)c * KP + kk] = 1.f;  // padding channels: X = 0 there
    }
  }
  cs = wave_sum(cs);
  sw = wave_sum(sw);
  if (a.pg_track) qw = wave_sum(qw);
  if (threadIdx.x == 0) {
    a.parts[wg] = cs;
    a.parts[nwg + wg] = sw;
    if (a.pg_track) a.parts[2 * nwg + wg] = qw;
  }
}

__global__ __launch_bounds__(256) void w_reduce_update_kernel(const WUpdateArgs a) {
  __shared__ double fscratch[5 * (ESPM_HP_NSCALAR + 2 * KP + 1)];
  __shared__ float s_part[8][32];
  __shared__ double s_rsw[4];
  const int nwg = a.k * a.nbk;
  if ((int)blockIdx.x >= nwg) {  // the extra workgroups: one value of the H-step's record reduction each (h_finalize_one)
#ifndef ESPM_EXPERIMENT_NO_FINALIZE   // (TIMING ONLY when defined: is the record reduction what this launch ends with?)
    h_finalize_one(a.fin, (int)blockIdx.x - nwg, fscratch);
#endif
    return;
  }
  const int kk = blockIdx.x / a.nbk, j = blockIdx.x - kk * a.nbk;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = 32 * j + col;
  const int e = kk * a.n_pad + c;
  auto src = [&](int b) { return reinterpret_cast<const float*>(a.src + (size_t)b * a.src_stride)[e]; };
  // Everything this workgroup needs from memory is requested up front - 32 sources per thread (256 slabs), its share of
  // the H-step's records - so that ONE memory round trip and ONE barrier separate the launch from the update: the kernel
  // is pure latency (8 us before, of an iteration of 150).
  constexpr int INFLIGHT = 32;
  float v[INFLIGHT];
  const bool live = c < a.n_pad;
#pragma unroll
  for (int u = 0; u < INFLIGHT; ++u) {
    const int b = grp + 8 * u;
    v[u] = (live && b < a.nsrc) ? src(b) : 0.f;
  }
  float wo_pre, fx_pre;
  w_update_preload(a, kk, c, threadIdx.x < 32, wo_pre, fx_pre);   // (lanes 0..31 of wave 0 own the 32 entries)
  double rsp = 0.0;   // slab mode: this thread's share of row sum kk of the new H (the order of h_finalize_body: same value as hstat's)
  if (a.hpart) {
    const size_t nb = a.nblk_h;
    for (int b = threadIdx.x; b < a.nblk_h; b += 256) rsp += a.hpart[(ESPM_HP_ROWSUM + kk) * nb + b];
  }
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = ((v[u] + v[u + 8]) + v[u + 16]) + v[u + 24];   // (the order of the loop below: source b ascending per partial)
  if (live) {
    int b = grp + 8 * INFLIGHT;
    for (; b + 56 < a.nsrc; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += src(b + 8 * u);
    }
    for (int u = 0; b < a.nsrc; b += 8, ++u) acc[u & 7] += src(b);
  }
  s_part[grp][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  // row sum of component kk of the new H
  double rs = 0.0;
  if (a.hpart) {
    rsp = wave_sum(rsp);
    if ((threadIdx.x & 63) == 0) s_rsw[threadIdx.x >> 6] = rsp;
    __syncthreads();   // (also orders s_part)
    rs = ((s_rsw[0] + s_rsw[1]) + s_rsw[2]) + s_rsw[3];   // wave order, like block_reduce
  } else if (a.hstat_rs) {
    rs = a.hstat_rs[ESPM_HS_ROWSUM + kk];
    __syncthreads();
  } else {
    for (int r = 0; r < a.nsrc; ++r)
      rs += reinterpret_cast<const double*>(a.src + (size_t)r * a.src_stride + a.rec_hstat_off)[ESPM_HS_ROWSUM + kk];
    if (blockIdx.x == 0 && threadIdx.x < ESPM_HS_STRIDE) {  // global statistics of the new H (as shard_combine)
      double t = 0.0;
      for (int r = 0; r < a.nsrc; ++r) {
        const double v2 = reinterpret_cast<const double*>(a.src + (size_t)r * a.src_stride + a.rec_hstat_off)[threadIdx.x];
        t = (int)threadIdx.x < ESPM_HS_MAX ? t + v2 : fmax(t, v2);
      }
      a.hstat_out[threadIdx.x] = t;
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) {  // wave 0; its first 32 lanes own the 32 entries
    float t = 0.f;
    const bool owns = grp == 0 && c < a.n_pad;
    if (owns) {
#pragma unroll
      for (int g = 0; g < 8; ++g) t += s_part[g][col];
    }
    w_update_entries(a, kk, c, e, owns, t, rs, nwg, blockIdx.x, wo_pre, fx_pre);
  }
}

// ---- W update under the simplex over W, G = identity, all rows in the simplex: many workgroups instead of one ----------------
// With G = identity every denominator of a column of W is the same number (rowsum of H', updates.py:60), so the multiplier's
// function is f(delta) = S / delta + n0 eps - 1 in delta = nu + rowsum, with S the sum of the positive numerators and n0 the
// rows without one: what the reference's bisection (dicotomy.py:111-173, global stop rule) does to it follows from S, the
// largest numerator and the counts alone.  w_reduce_kernel leaves those per 32 channels (WReduceArgs::bparts); here EVERY
// workgroup (component kk, 32 channels - the geometry of w_reduce_update_kernel) adds the partials of all components in a
// fixed order, walks the same decisions as w_finish_fast_kernel - bracket, root, the sweep the reference stops at (estimate
// from the linearisation, the exact f where that is within 1 % of the tolerance), the midpoint of that sweep - and updates
// its 32 entries with delta in the place of the row sum.  37 us of one workgroup become 8 us of 320.
// (A positive numerator below eps delta counts as itself, not as eps: at most n eps = 2e-11 of f.)
struct WSimplexArgs {
  WUpdateArgs u;             // the update of the entries and what it leaves for the tail (src / hpart unused: A is read from u.a_out)
  const double* bparts;      // [k * nbk][3]
  const double* hstat;       // statistics of the new H (row sums)
  double rows;               // rows of W (all of them in the simplex)
  double tol;
};

__global__ __launch_bounds__(64) void w_simplex_update_kernel(const WSimplexArgs x) {
  const WUpdateArgs& a = x.u;
  const int nwg = a.k * a.nbk;
  const int kk = blockIdx.x / a.nbk, j = blockIdx.x - kk * a.nbk;
  const int lane = threadIdx.x;
  // the entries first: their loads fly while the multipliers are worked out
  const int c = 32 * j + (lane & 31);
  const int e = kk * a.n_pad + c;
  const bool owns = lane < 32 && c < a.n_pad;
  const float t_e = owns ? a.a_out[e] : 0.f;
  float wo_pre, fx_pre;
  w_update_preload(a, kk, c, owns, wo_pre, fx_pre);
  double ssum[KP], smax[KP], spos[KP], rs[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) {
    ssum[q] = 0.0; smax[q] = 0.0; spos[q] = 0.0; rs[q] = 1.0;
    if (q < a.k) {
      double s1 = 0.0, s3 = 0.0, s2 = 0.0;
      for (int b = lane; b < a.nbk; b += 64) {
        const double* p = x.bparts + 3 * ((size_t)q * a.nbk + b);
        s1 += p[0];
        s2 = fmax(s2, p[1]);
        s3 += p[2];
      }
      ssum[q] = wave_sum(s1);
      smax[q] = wave_max(s2);
      spos[q] = wave_sum(s3);
      rs[q] = x.hstat[ESPM_HS_ROWSUM + q];
    }
  }
  // per component (every lane the same arithmetic): bracket [ad, ad + width] in delta, root, slope, place of the root
  const double eps = (double)a.log_shift;
  double root[KP], fder[KP], ad[KP], width[KP], uu[KP], cst[KP];
  bool solve[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) {
    solve[q] = q < a.k && ssum[q] > 0.0 && ssum[q] < INFINITY;
    const double den = (double)(float)rs[q];                 // dv = colsum(G) * (float) rowsum = the fp32 row sum
    const double lo = smax[q] / 2 - den;                     // a, dicotomy.py:29-43
    const double hi = x.rows * smax[q] / 0.5 - den;          // b, dicotomy.py:49
    cst[q] = (x.rows - spos[q]) * eps;                       // the rows without a positive numerator: eps each
    ad[q] = lo + den;
    width[q] = hi - lo;
    // Newton from delta = S: f(S) = cst; the finish accepts |f| <= 1e-11, else converges to S / (1 - cst)
    root[q] = solve[q] ? (fabs(cst[q]) <= 1e-11 ? ssum[q] : ssum[q] / (1.0 - cst[q])) : 1.0;
    fder[q] = solve[q] ? -ssum[q] / (root[q] * root[q]) : 0.0;
    uu[q] = solve[q] ? fmin(fmax((root[q] - ad[q]) / width[q], 0.0), 1.0) : 0.5;
  }
  auto mid_frac = [](double u1, int t) {
    const double scale = ldexp(1.0, t - 1);
    const double cell = fmin(floor(u1 * scale), scale - 1.0);
    return ldexp(2.0 * cell + 1.0, -t);
  };
  // the sweep the reference stops at: lane l looks at sweeps l + 1 and l + 65 (at most 101)
  unsigned long long mc[2], mb[2];
  for (int hf = 0; hf < 2; ++hf) {
    const int t = 1 + lane + 64 * hf;
    double est = 0.0;
#pragma unroll
    for (int q = 0; q < KP; ++q)
      if (q < a.k) est = fmax(est, fabs(fder[q] * ((ad[q] + width[q] * mid_frac(uu[q], t)) - root[q])));
    const bool certain = t <= 101 && (est <= 0.99 * x.tol || t == 101);
    const bool band = t <= 101 && !certain && est <= 1.01 * x.tol;
    mc[hf] = __ballot(certain);
    mb[hf] = __ballot(band);
  }
  int t_stop = 101;
  for (;;) {   // (uniform)
    const unsigned long long w0 = mc[0] | mb[0], w1 = mc[1] | mb[1];
    if (!w0 && !w1) break;
    const int hf = w0 ? 0 : 1;
    const int bit = __ffsll((long long)(hf ? w1 : w0)) - 1;
    const int t = 1 + bit + 64 * hf;
    if ((mc[hf] >> bit) & 1ull) { t_stop = t; break; }
    double worst = 0.0;
#pragma unroll
    for (int q = 0; q < KP; ++q)
      if (q < a.k && fder[q] != 0.0) {
        const double d = ad[q] + width[q] * mid_frac(uu[q], t);
        worst = fmax(worst, fabs(ssum[q] / d + cst[q] - 1.0));
      }
    if (worst <= x.tol) { t_stop = t; break; }
    mb[hf] &= ~(1ull << bit);
  }
  // delta of this workgroup's component: den + nu = (den - d*) + delta = delta (no multiplier without a positive numerator)
  double delta = rs[0];
#pragma unroll
  for (int q = 0; q < KP; ++q)
    if (q == kk) delta = solve[q] ? ad[q] + width[q] * mid_frac(uu[q], t_stop) : (double)(float)rs[q];
  w_update_entries(a, kk, c, e, owns, t_e, delta, nwg, blockIdx.x, wo_pre, fx_pre);
}

// ---- Slab reduction, record exchange and W update of a SHARDED image in ONE launch (espm_mu_shard_exchange_finish) ----------
// The sharded W step after the accumulation used to be four launches - slab reduction into the rank's record, post, wait,
// sum over the ranks' records + update - i.e. ~17 us of launch latencies around 100 KB of traffic.  Here the reduction
// workgroup (kk, j) that has summed the slabs for ITS 32 entries of A delivers that 128-byte piece to every rank's mailbox
// itself (P2P stores), raises ITS flag there, waits for the same piece of every other rank, sums the pieces in rank order
// (bit-identical W on all ranks) and updates its entries of W: the exchange is as fine-grained as the dependency.  The extra
// workgroup reduces the H-step's records, delivers the statistics of the new H block to every rank and its boundary rows
// to the neighbours (and to itself), and raises flag `nwg`; every reduction workgroup also waits for that flag of every
// rank (the global row sums are in the denominator of W'), so by the end of the launch the halo rows of the next H-step
// have arrived too.  Every workgroup posts before it waits and the grid (k n_pad / 32 + 1 workgroups of 256 threads) is
// resident at once: no deadlock; waits are bounded (a lost peer is counted in the mailbox's error word).
#ifndef ESPM_XCHG_SMALL_WORLD
#define ESPM_XCHG_SMALL_WORLD 8
#endif
struct WExchangeArgs {
  WUpdateArgs u;
  unsigned char* mbox[16];   // every rank's mailbox as mapped here
  int world, rank, nfl, with_halo;
  size_t rec_bytes, slot_base, wgflags_off, gran_off, err_off, hstat_off, top_off, bot_off;   // (gran_off: of this sequence number's parity)
  unsigned int seq;
  long long max_ticks;
  int release;               // flags stored with release order at system scope (espm_xchg_set_order) instead of behind s_waitcnt vmcnt(0) alone
  const float* halo_h;
  int halo_k, halo_nx, halo_ny, halo_ppad;
  // simplex over W with G = identity: the summed A and, per reduction workgroup, what the bracket of its component's multiplier
  // needs (the three doubles w_reduce_kernel leaves) go out INSTEAD of the update - w_simplex_update_kernel follows; else null
  double* bparts;
};

// The flag behind a record's stores (the ordering contract: mu_xchg.hip).  Default: a relaxed system-scope store - the data stores were
// write-through, every storing wave drained them (s_waitcnt vmcnt(0)) and the workgroup's barrier lies in between.  release != 0
// (ESPM_XCHG_ORDER=release, espm_xchg_set_order): the same store with release order at system scope - the compiler's full recipe
// (write back this XCD's L2, wait, store), by the flag-storing lanes only.
__device__ __forceinline__ void xchg_store_flag(unsigned int* flag, unsigned int seq, int release) {
  if (release) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void xchg_wait_flag(const unsigned int* flag, unsigned int seq, long long max_ticks, unsigned int* err) {
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {   // (polling with acquire loads would invalidate the caches per poll)
    if (wall_clock64() - t0 > max_ticks) {
      atomicAdd(err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int MAXW>   // ranks the unrolled polls are laid out for (8: a node; 16: the mailbox's limit) - the kernel's code size follows it
__global__ __launch_bounds__(256) void w_exchange_update_kernel(const WExchangeArgs x) {
  const WUpdateArgs& a = x.u;
  __shared__ double fscratch[5 * (ESPM_HP_NSCALAR + 2 * KP + 1)];
  __shared__ float s_part[8][32];
  const int nwg = a.k * a.nbk;
  auto record = [&](int dst, int src_rank) { return x.mbox[dst] + x.slot_base + (size_t)src_rank * x.rec_bytes; };
  auto flag = [&](int dst, int src_rank, int idx) {
    return reinterpret_cast<unsigned int*>(x.mbox[dst] + x.wgflags_off + ((size_t)src_rank * x.nfl + idx) * sizeof(unsigned int));
  };
  // Progress without the whole grid being resident (grids of k n_pad / 32 + 1 workgroups exceed what the device
  // holds at once from k = 8, n = 8192 on): workgroups are dispatched in index order, every workgroup POSTS before it waits,
  // and what it waits for was posted by the workgroup of the SAME index on every rank (reduction workgroup wg: the pieces of
  // the workgroups wg) or by the first H_FINALIZE_JOBS workgroups (the record reduction's: only reduction workgroup 0 waits for them).  The
  // lowest-indexed unfinished workgroup of every rank is therefore resident and can finish; whatever it frees lets the next
  // one in.  (Round 2 had the extra workgroup LAST and every reduction workgroup waiting for it: a grid beyond the residency
  // dead-locked for the 2 s bound of the waits.  An occupancy query as a second guard was tried and dropped: on this stack
  // hipOccupancyMaxActiveBlocksPerMultiprocessor answers 2 workgroups per CU for this 256-thread kernel, which would send
  // the headline's 321 workgroups down the four-launch path - profiles/r03c_shard_iter_residency_guard.log.)  The waits stay bounded.
  // granule g of rank src_rank's contribution, in rank dst's mailbox (GRAN granules per source rank): g = 32 wg + column for a
  // reduction workgroup's piece, 32 nfl + 2 wg + half for its row sum, 34 nfl + 2 i + half for statistic i of the new H block
  const size_t GRAN = (size_t)34 * x.nfl + 2 * ESPM_HS_STRIDE;
  auto gran = [&](int dst, int src_rank, int g) {
    return reinterpret_cast<unsigned long long*>(x.mbox[dst] + x.gran_off) + (size_t)src_rank * GRAN + g;
  };
  constexpr int NJ = H_FINALIZE_JOBS;
  if ((int)blockIdx.x < NJ) {  // the extra workgroups, FIRST in the grid: one value of the record reduction each (h_finalize_one);
    // the one whose value needs no records (SUMY) also posts the boundary rows of this rank's new H block
    const int job = (int)blockIdx.x;
    const bool halo_job = job == H_FINALIZE_NV;
    // the boundary rows first - they are in memory since the launch before this one, and nobody needs them before the NEXT
    // launch: their stores are long acknowledged when the flag that covers them is raised at the end
    // (every store into a mailbox is a system-scope write-through store: whatever memory type a peer's mapping has here,
    //  the data is on its way to that rank's memory when the store is acknowledged, and `s_waitcnt vmcnt(0)` orders the flag)
    if (halo_job && x.with_halo) {
      for (int d = -1; d <= 1; ++d) {   // the neighbours read these rows as their halo; this rank keeps a copy (record layout)
        const int r = x.rank + d;
        if (r < 0 || r >= x.world) continue;
        float* top = reinterpret_cast<float*>(record(r, x.rank) + x.top_off);
        float* bot = reinterpret_cast<float*>(record(r, x.rank) + x.bot_off);
        if ((x.halo_ny & 3) == 0) {   // rows of whole quads: 16-byte write-through stores (a 4-byte store to a peer is a fabric write of its own)
          typedef float xf4 __attribute__((ext_vector_type(4)));
          for (int e = 4 * threadIdx.x; e < x.halo_k * x.halo_ny; e += 4 * 256) {
            const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
            const xf4 vt = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + j);
            const xf4 vb = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(top + e), "v"(vt) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(bot + e), "v"(vb) : "memory");
          }
        } else {
          for (int e = threadIdx.x; e < x.halo_k * x.halo_ny; e += 256) {
            const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
            __hip_atomic_store(top + e, x.halo_h[(size_t)kk * x.halo_ppad + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(bot + e, x.halo_h[(size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
    // a statistic: reduced, then one granule {half of a double, sequence number} per half and destination - value and "it is
    // there" in one store, like the pieces: reduction workgroup 0 of every rank has it one trip over the link after the reduction
    __shared__ double s_hstat[ESPM_HS_STRIDE];
    HFinalizeArgs fin = a.fin;
    double* rec_hstat = fin.hstat_out;   // this rank's record in its OWN mailbox (the plain copy: espm_xchg_records' readers)
    fin.hstat_out = s_hstat;
    h_finalize_one(fin, job, fscratch);
    if (halo_job) {
      // the boundary rows' stores were issued long ago: this wait returns at once
      // (the mailboxes are uncached memory: a store is delivered once it is acknowledged - no cache to write back, so no
      //  system-scope fence, which would flush this XCD's whole L2 - only ORDER: every thread's stores before any flag)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if ((int)threadIdx.x < x.world) xchg_store_flag(flag(threadIdx.x, x.rank, nwg), x.seq, x.release);
      return;
    }
    // which entry of the statistics this job's value is (row sums [0, KP), maxima [KP, 2 KP)); the others go to the history only
    int js = -1;
    if (job >= ESPM_HP_ROWSUM && job < ESPM_HP_ROWSUM + KP) js = ESPM_HS_ROWSUM + (job - ESPM_HP_ROWSUM);
    if (job >= 5 + KP && job < 5 + 2 * KP) js = ESPM_HS_MAX + (job - (5 + KP));
    if (js < 0) return;
    __syncthreads();   // (thread 0 of h_finalize_one wrote s_hstat[js])
    if ((int)threadIdx.x < 2 * x.world) {
      const int r = threadIdx.x >> 1, half_i = threadIdx.x & 1;
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, s_hstat[js]);
      const unsigned int half = half_i ? (unsigned int)(bits >> 32) : (unsigned int)bits;
      __hip_atomic_store(gran(r, x.rank, 34 * x.nfl + 2 * js + half_i), ((unsigned long long)x.seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) rec_hstat[js] = s_hstat[js];
    return;
  }
  const int wg = (int)blockIdx.x - NJ;   // reduction workgroup (component kk, 32 channels)
  const int kk = wg / a.nbk, j = wg - kk * a.nbk;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = 32 * j + col;
  const int e = kk * a.n_pad + c;
  auto src = [&](int b) { return reinterpret_cast<const float*>(a.src + (size_t)b * a.src_stride)[e]; };
  constexpr int INFLIGHT = 32;
  float v[INFLIGHT];
  const bool live = c < a.n_pad;
#pragma unroll
  for (int u = 0; u < INFLIGHT; ++u) {
    const int b = grp + 8 * u;
    v[u] = (live && b < a.nsrc) ? src(b) : 0.f;
  }
  float wo_pre, fx_pre;
  w_update_preload(a, kk, c, threadIdx.x < 32, wo_pre, fx_pre);
  // Row sum kk of THIS rank's new H block from the H-step's records, in the order of h_finalize_body (the same value the extra
  // workgroup leaves in the statistics): it travels with the piece, so that the W update waits for pieces only - not for the
  // extra workgroups' record reductions, which only workgroup 0 (the global statistics for the NEXT H-step) still waits for.
  __shared__ double s_rsw[4];
  double rsp = 0.0;
  {
    const size_t nb = a.nblk_h;
    for (int b = threadIdx.x; b < a.nblk_h; b += 256) rsp += a.hpart[(ESPM_HP_ROWSUM + kk) * nb + b];
  }
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = ((v[u] + v[u + 8]) + v[u + 16]) + v[u + 24];
  if (live) {
    int b = grp + 8 * INFLIGHT;
    for (; b + 56 < a.nsrc; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += src(b + 8 * u);
    }
    for (int u = 0; b < a.nsrc; b += 8, ++u) acc[u & 7] += src(b);
  }
  s_part[grp][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  rsp = wave_sum(rsp);
  if ((threadIdx.x & 63) == 0) s_rsw[threadIdx.x >> 6] = rsp;
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0: lanes 0..31 own the 32 entries, lanes 32 and 33 the two halves of the row sum
    const int lane = threadIdx.x;
    const bool owns = grp == 0 && live;
    float t = 0.f;
    if (owns) {
#pragma unroll
      for (int g = 0; g < 8; ++g) t += s_part[g][col];
    }
    // A value travels as ONE 8-byte granule {bits, sequence number} (system scope, write-through): the store that delivers it is
    // also what says it is there - no drain, no flag, no second trip over the link (MI355X_MICROARCH.md, handoff-1to1 against
    // handoff-flag; an aligned 8-byte store is observed whole).  The slot of a sequence number's parity is rewritten two
    // exchanges later, which a peer can only do after this rank posted the exchange in between (mu_xchg.hip).
    const double rs_mine = ((s_rsw[0] + s_rsw[1]) + s_rsw[2]) + s_rsw[3];   // wave order, like block_reduce
    const unsigned long long rs_bits = __builtin_bit_cast(unsigned long long, rs_mine);
    const bool sends = owns || lane == 32 || lane == 33;
    const int g_idx = lane < 32 ? 32 * wg + col : 32 * x.nfl + 2 * wg + (lane - 32);
    const unsigned int bits = lane < 32 ? __float_as_uint(t) : (lane == 32 ? (unsigned int)rs_bits : (unsigned int)(rs_bits >> 32));
    const unsigned long long mine = ((unsigned long long)x.seq << 32) | bits;
    if (sends)
      for (int r = 0; r < x.world; ++r) __hip_atomic_store(gran(r, x.rank, g_idx), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // every rank's granule of the same index: polled together (one load per rank in flight), bounded
    unsigned int* err = reinterpret_cast<unsigned int*>(x.mbox[x.rank] + x.err_off);
    unsigned int got[MAXW];
    {
      const long long t0 = wall_clock64();
      bool all = !sends;
      for (;;) {
        unsigned long long v[MAXW];
#pragma unroll
        for (int r = 0; r < MAXW; ++r)
          v[r] = (sends && r < x.world) ? __hip_atomic_load(gran(x.rank, r, g_idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ((unsigned long long)x.seq << 32);
        all = true;
#pragma unroll
        for (int r = 0; r < MAXW; ++r) {
          all = all && (unsigned int)(v[r] >> 32) == x.seq;
          got[r] = (unsigned int)v[r];
        }
        if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
        if (wall_clock64() - t0 > x.max_ticks) {   // a peer that never delivers must not hang the device: give up, count it
          if (!all) atomicAdd(err, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // added in rank order: the same sums on every rank
    float tt = 0.f;
    double rs = 0.0;
#pragma unroll
    for (int r = 0; r < MAXW; ++r)
      if (r < x.world) {
        tt += __uint_as_float(got[r]);
        const unsigned long long lo = (unsigned int)__builtin_amdgcn_readlane((int)got[r], 32), hi = (unsigned int)__builtin_amdgcn_readlane((int)got[r], 33);
        rs += __builtin_bit_cast(double, (hi << 32) | lo);
      }
    if (x.bparts) {   // the numerators' sum, maximum and count of positives (dicotomy.py:29-49), as w_reduce_kernel forms them
      if (owns) a.a_out[e] = tt;
      const float num = (owns && c < a.n) ? wo_pre * tt : 0.f;   // updates.py:59 (G = identity)
      double sum = num > 0.f ? (double)num : 0.0, cnt = num > 0.f ? 1.0 : 0.0;
      float mx = fmaxf(num, 0.f);
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      }
      if (lane == 0) {
        x.bparts[3 * (size_t)wg] = sum;
        x.bparts[3 * (size_t)wg + 1] = (double)mx;
        x.bparts[3 * (size_t)wg + 2] = cnt;
      }
    } else {
      w_update_entries(a, kk, c, e, owns, tt, rs, nwg, wg, wo_pre, fx_pre);
    }
#ifdef ESPM_EXPERIMENT_XCHG_NO_STATS_WAIT   // TIMING ONLY: is the statistics' path (record reduction in the extra workgroup, granules, this wait) what the launch ends with?
    if (false) {
#else
    if (wg == 0) {  // global statistics of the new H (as shard_combine): every rank's extra workgroup sends them as granules
#endif
      // one lane per half of a statistic: 2 ESPM_HS_STRIDE halves, 64 per pass (one pass up to 16 components, two in the widest build)
      for (int hl = lane; hl < ((2 * ESPM_HS_STRIDE + 63) & ~63); hl += 64) {
      const bool polls = hl < 2 * ESPM_HS_STRIDE;
      unsigned int hv[MAXW];
      {
        const long long t0 = wall_clock64();
        bool all = !polls;
        for (;;) {
          unsigned long long v[MAXW];
#pragma unroll
          for (int r = 0; r < MAXW; ++r)
            v[r] = (polls && r < x.world) ? __hip_atomic_load(gran(x.rank, r, 34 * x.nfl + hl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                          : ((unsigned long long)x.seq << 32);
          all = true;
#pragma unroll
          for (int r = 0; r < MAXW; ++r) {
            all = all && (unsigned int)(v[r] >> 32) == x.seq;
            hv[r] = (unsigned int)v[r];
          }
          if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
          if (wall_clock64() - t0 > x.max_ticks) {
            if (!all) atomicAdd(err, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      double g = 0.0;
#pragma unroll
      for (int r = 0; r < MAXW; ++r)
        if (r < x.world) {   // lane 2 i holds the low half of statistic i, lane 2 i + 1 the high half
          const unsigned int other = (unsigned int)__shfl_xor((int)hv[r], 1, 64);
          const unsigned long long bits = (lane & 1) ? (((unsigned long long)hv[r] << 32) | other) : (((unsigned long long)other << 32) | hv[r]);
          const double v2 = __builtin_bit_cast(double, bits);
          g = (hl >> 1) < ESPM_HS_MAX ? g + v2 : fmax(g, v2);
        }
      if (polls && !(lane & 1)) a.hstat_out[hl >> 1] = g;
      }
      // the boundary rows of the neighbours (the next launch reads them): their flag
      if (lane < x.world) xchg_wait_flag(flag(x.rank, lane, nwg), x.seq, x.max_ticks, err);
    }
  }
}

// Column sums of G W' and rel_W (base.py:323) from the partials and W', W: one workgroup (w_tail_body, mu_common.hpp).
// 256 threads: the cross-wave stage of a block reduction costs per wave, and everything here is latency - the
// entries of W are requested up front, before the partials are reduced, so that only one memory round trip and
// two short reductions separate the launch from the result.
constexpr int WT_THREADS = 256;
__global__ __launch_bounds__(WT_THREADS) void w_update_tail_kernel(const WTailArgs a) {
  __shared__ double scratch[(WT_THREADS / 64 + 1) * (KP + 1) + 1];
  w_tail_body<40>(a, scratch);   // 40 x 256 = 10240 entries of W held in registers (the headline size); more take the loop
}

// ---- W finish: one workgroup of 1024 threads --------------------------------------------------

__device__ __forceinline__ double block_sum1(double v, double* scratch) {
  double a[1] = {v};
  block_reduce<1, 1>(a, scratch);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = a[0];
  __syncthreads();
  return bc;
}
__device__ __forceinline__ double block_max1(double v, double* scratch) {
  double a[1] = {v};
  block_reduce<1, 0>(a, scratch);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = a[0];
  __syncthreads();
  return bc;
}

// Register-resident variant for M <= WF_ROWS * 1024 rows (and M * k <= WF_GTA_MAX when G is given): thread t
// owns rows t, t + 1024, ... of W and keeps their old / numerator / denominator / new entries in registers
// across the phases (no index division, no scratch traffic); the second stage of the slab reduction
// (sum over the `nsplit` partials, fixed order) is folded into the load of A; G^T A is formed by one wave
// per output entry and G W' reads W' from LDS.  Same arithmetic and the same global-stop bisection as
// w_finish_kernel below; only the data movement differs.
constexpr int WF_GTA_MAX = 8192;
constexpr int WF_GTA_PAR = 512;  // M * k up to which G^T A uses the all-threads path (16 x M k floats of LDS)

// One step of a packed butterfly sum over the lanes of a wave: of the first N values a lane keeps one half
// (the upper one when `up`) and adds the partner's copy of that half, which the partner sends instead of keeping.
template <int N>
__device__ __forceinline__ void butterfly_half(float (&v)[32], bool up, int off) {
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const float send = up ? v[j] : v[N / 2 + j];
    const float keep = up ? v[N / 2 + j] : v[j];
    v[j] = keep + __shfl_xor(send, off, 64);
  }
}

__device__ __forceinline__ float load_a(const WFinishArgs& a, int kk, int c) {
  return a.a[(size_t)kk * a.n_pad + c];
}

// 1 / x for the bisection's sum_c num_c / (nu + den_c): v_rcp_f64 and two Newton steps (a few ulp; an IEEE division is
// three times the instructions, and 10 of them per thread and step were what the one-workgroup W finish spent its time on)
__device__ __forceinline__ double rcp_f64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// NT threads: 1024 (16 waves, 128 registers each) or 512 (8 waves, 256 registers): the thread count that spills less wins -
// with G = identity and the simplex over W the 16-wave version kept 80 of its values in scratch memory and took 49 us, the
// 8-wave one takes 37 (tools/analysis/w_finish_clock.py); with a dictionary G the loops over its rows want the 16 waves.
// WF_ROWS rows of W per thread (its state stays in registers through the phases); CROWS channels per thread in the phase that
// forms G^T A with thread = channel (a dictionary G has few rows - W state for ONE row per thread - and many channels: sizing
// the W state by the channels spilled 245 registers at 8 components).
template <int KK, int WF_ROWS, int NT, int CROWS = WF_ROWS>
__global__ __launch_bounds__(NT) void w_finish_fast_kernel(const WFinishArgs a) {
  constexpr int KA = KK;  // per-thread arrays are sized by the real component count (k == KK)
  __shared__ double scratch[(NT / 64 + 1) * KP];
  __shared__ double bis[2][(NT / 64) * 2 * KA];   // per-wave partial sums (f, f') of the root finder, two alternating buffers
  __shared__ double s_lo[KA], s_hi[KA], s_mid[KA], s_dstar[KA], s_sum[KA];
  __shared__ double s_x[KA], s_root[KA], s_fder[KA], s_ad[KA], s_width[KA], s_u[KA];
  __shared__ int s_flag[KA];
  __shared__ unsigned long long s_mask[4];
  __shared__ int s_go;
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // G given: [M*k] new W, [M*k] G^T A
  const int M = a.m > 0 ? a.m : a.n;
  const int k = a.k;
  const int MK = M * k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int SW = (KA + 3) / 4 * 4;   // stride of a row of W' in LDS: whole 16-byte quads (the rows of G W' read them as float4)
  float* s_w = dyn;
  float* s_gta = dyn + (size_t)M * SW;
  ESPM_PHASE_STAMP(0);
#ifdef ESPM_PHASE_CLOCK
  if (threadIdx.x == 0 && espm_phase_buf) espm_phase_buf[20] = (unsigned long long)clock64();   // shader clock ticks (against the 100 MHz stamps)
#endif

  float wn[WF_ROWS][KA];
#pragma unroll
  for (int r = 0; r < WF_ROWS; ++r)
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) wn[r][kk] = 0.f;

  if (a.update_w) {
    if (KA <= 8 && a.g && a.g_t && MK <= WF_GTA_PAR) {   // (the packed sum below is laid out for 4 rows x 8 components)
      // G^T A with the association G^T (R H^T), updates.py:58-59, with every thread busy: thread = channel
      // (its k entries of A in registers), G^T rows read coalesced, the products of a wave summed across its lanes
      // and the per-wave partials by the workgroup.  s_part lives behind s_gta.
      float* s_part = s_gta + MK;  // [NT / 64][MK]
      float av[CROWS][KA];
#pragma unroll
      for (int r = 0; r < CROWS; ++r) {
        const int c = tid + r * NT;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) av[r][kk] = (kk < k && c < a.n) ? load_a(a, kk, c) : 0.f;
      }
      // Batches of 4 rows of G^T: 32 (row, component) products per lane, summed over the 64 lanes by a packed
      // butterfly - in every step a lane hands the half of its values it does not keep to its partner - which
      // takes 32 cross-lane moves instead of 6 x 32 (the moves, on one CU, are what bounds this phase).
      constexpr int MB = 4, KB8 = 8;
      for (int m0 = 0; m0 < a.m; m0 += MB) {
        float gv[MB][CROWS];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          const int mm = m0 + b < a.m ? m0 + b : a.m - 1;
#pragma unroll
          for (int r = 0; r < CROWS; ++r) {
            const int c = tid + r * NT;
            gv[b][r] = c < a.n ? a.g_t[(size_t)mm * a.n_pad + c] : 0.f;
          }
        }
        float v[MB * KB8];
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
          for (int kk = 0; kk < KB8; ++kk) {
            float t = 0.f;
            if (kk < KA) {
#pragma unroll
              for (int r = 0; r < CROWS; ++r) t = fmaf(gv[b][r], av[r][kk < KA ? kk : 0], t);
            }
            v[b * KB8 + kk] = t;
          }
        butterfly_half<32>(v, (lane & 32) != 0, 32);
        butterfly_half<16>(v, (lane & 16) != 0, 16);
        butterfly_half<8>(v, (lane & 8) != 0, 8);
        butterfly_half<4>(v, (lane & 4) != 0, 4);
        butterfly_half<2>(v, (lane & 2) != 0, 2);
        const float total = v[0] + __shfl_xor(v[0], 1, 64);
        const int idx = (lane >> 1) & 31, b = idx >> 3, kk = idx & 7;  // the (row, component) this lane pair ended up with
        if (!(lane & 1) && m0 + b < a.m && kk < k) s_part[wave * MK + (m0 + b) * k + kk] = total;
      }
      __syncthreads();
      for (int o = tid; o < MK; o += NT) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += s_part[w * MK + o];
        s_gta[o] = t;
      }
      __syncthreads();
    } else if (a.g) {  // no transposed copy: one wave per (mm, kk), G read with a stride of m
      for (int o = wave; o < MK; o += NT / 64) {
        const int mm = o / k, kk = o - mm * k;
        float sacc = 0.f;
        for (int c = lane; c < a.n; c += 64) sacc = fmaf(a.g[(size_t)c * a.m + mm], load_a(a, kk, c), sacc);
        sacc = wave_sum(sacc);
        if (lane == 0) s_gta[o] = sacc;
      }
      __syncthreads();
    }
    ESPM_PHASE_STAMP(1);   // (instrumented build, tools/analysis/w_finish_clock.py) G^T A done
    float rs[KA];
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) rs[kk] = kk < k ? (float)a.hstat[ESPM_HS_ROWSUM + kk] : 0.f;
    float wo[WF_ROWS][KA], nv[WF_ROWS][KA], dv[WF_ROWS][KA], pgrad[WF_ROWS][KA];
    bool in_set[WF_ROWS];
#pragma unroll
    for (int r = 0; r < WF_ROWS; ++r) {
      const int mm = tid + r * NT;
      in_set[r] = false;
#pragma unroll
      for (int kk = 0; kk < KA; ++kk) { wo[r][kk] = 0.f; nv[r][kk] = 0.f; dv[r][kk] = 1.f; pgrad[r][kk] = 0.f; }
      if (mm < M) {
        in_set[r] = !a.simplex_rows || a.simplex_rows[mm];
        const float cg = a.g ? a.colsum_g[mm] : 1.f;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
          if (kk < k) {
            const float gta = a.g ? s_gta[mm * k + kk] : load_a(a, kk, mm);
            wo[r][kk] = a.w_old[mm * k + kk];
            nv[r][kk] = wo[r][kk] * gta;        // updates.py:59
            dv[r][kk] = cg * rs[kk];            // updates.py:60
            if (a.pg_gamma_w > 0.f) {           // projected gradient: W - (colsum(G) rowsum(H) - G^T A) / gamma, updates.py:353-362
              pgrad[r][kk] = dv[r][kk] - gta;
              nv[r][kk] = wo[r][kk] - pgrad[r][kk] / a.pg_gamma_w;
              dv[r][kk] = 1.f;
            } else if (a.breg_sr) {             // Bregman variant (G = identity), updates.py:41-48
              const float sr = a.xscale * a.breg_sr[mm];
              dv[r][kk] = (dv[r][kk] - gta) * wo[r][kk] + sr;
              nv[r][kk] = sr * wo[r][kk];
            }
          }
        }
      }
    }
    ESPM_PHASE_STAMP(10);   // loads, numerators, denominators
    if (a.simplex_w) {
      // Multipliers of the simplex over W, all components at once.  The reference bisects the bracket [a, b] of every
      // column with a GLOBAL stop rule (dicotomy.py:146-171): all columns stop at the first sweep t in which every column's
      // midpoint has |f| <= tol, so its nu is the t-th midpoint of the bisection path towards the root - accurate to tol
      // only, and W' inherits that (column sums 1 +- 1e-5).  Walking those ~40 sweeps on one CU was 90 us of a 260 us
      // iteration.  The same nu in ~5 evaluations of f instead of ~40:
      //   1. the ROOT by Newton in the shifted unknown delta = nu + d* (as the H-step's simplex_root): f is convex and
      //      decreasing, sum(num) bounds the root from the right (every e = den - d* >= 0), so the first step lands left of
      //      the root and the rest converges monotonically; safeguarded by the running bracket;
      //   2. the bisection path needs no sweeps once the root is known: the t-th midpoint towards a root at fraction u of
      //      the bracket is a + (b - a) (2 floor(u 2^(t-1)) + 1) / 2^t;
      //   3. the sweep the reference stops at: the first t with max_k |f_k(mid_t)| <= tol - decided from the linearisation
      //      f ~ f'(root) (mid - root) (its relative error is |mid - root| / delta ~ tol there), and by a real evaluation of
      //      f where the estimate is within 1 % of tol.
      // Thread kk < k OWNS component kk (bracket, iterate, decisions); the evaluation points travel through LDS (two
      // barriers per evaluation): per-thread copies of the state of all components cost more registers than a wave has.
      double cnt_l = 0.0;
#pragma unroll
      for (int r = 0; r < WF_ROWS; ++r) cnt_l += (tid + r * NT < M && in_set[r]) ? 1.0 : 0.0;
      const double rows = block_sum1(cnt_l, scratch);
      {  // per column: max(num/2 - den), max num, max(-den) (dicotomy.py:29-49); max(-den | num > 0), sum num
        double b1[KA], b2[KA], b3[KA];
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) { b1[kk] = -INFINITY; b2[kk] = 0.0; b3[kk] = -INFINITY; }
#pragma unroll
        for (int r = 0; r < WF_ROWS; ++r) {
          if (tid + r * NT < M && in_set[r]) {
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
              if (kk < k) {
                const double nn = nv[r][kk], dd = dv[r][kk];
                if (nn > 0) b1[kk] = fmax(b1[kk], nn / 2 - dd);
                b2[kk] = fmax(b2[kk], nn);
                b3[kk] = fmax(b3[kk], -dd);
              }
            }
          }
        }
        block_reduce<KA, 0>(b1, scratch);
        block_reduce<KA, 0>(b2, scratch);
        block_reduce<KA, 0>(b3, scratch);
        if (tid == 0)
          for (int kk = 0; kk < k; ++kk) {
            s_lo[kk] = b1[kk];                          // a, dicotomy.py:29-43
            s_hi[kk] = rows * b2[kk] / 0.5 + b3[kk];    // b, dicotomy.py:49
          }
      }
      {
        double b4[KA], b5[KA];
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) { b4[kk] = -INFINITY; b5[kk] = 0.0; }
#pragma unroll
        for (int r = 0; r < WF_ROWS; ++r) {
          if (tid + r * NT < M && in_set[r]) {
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
              if (kk < k && nv[r][kk] > 0.f) {
                b4[kk] = fmax(b4[kk], -(double)dv[r][kk]);
                b5[kk] += (double)nv[r][kk];
              }
            }
          }
        }
        block_reduce<KA, 0>(b4, scratch);
        block_reduce<KA, KA>(b5, scratch);
        if (tid == 0)
          for (int kk = 0; kk < k; ++kk) {
            s_dstar[kk] = b5[kk] > 0.0 ? -b4[kk] : 0.0;   // d* = min{den : num > 0}: the last pole of f is at nu = -d*
            s_sum[kk] = b5[kk];
          }
      }
      __syncthreads();
      constexpr int NWV = NT / 64;
      const double tol = (double)a.tol;
      // owner state (threads kk < k); a column without a positive numerator has no multiplier (dicotomy.py:19)
      const bool owner = tid < k;
      const bool solve = owner && s_sum[owner ? tid : 0] > 0.0 && s_sum[owner ? tid : 0] < INFINITY;
      double o_x = 1.0, o_lo = 0.0, o_hi = 1.0, o_dxold = 1.0, o_fder = 0.0;
      bool o_done = true;
      if (owner) {
        const double dstar = s_dstar[tid];
        if (solve) {
          o_lo = fmax(s_lo[tid] + dstar, 0.0);
          o_hi = s_hi[tid] + dstar;
          o_x = fmin(fmax(s_sum[tid], o_lo), o_hi);
          o_dxold = o_hi - o_lo;
          o_done = false;
        }
        s_x[tid] = o_x;
        s_flag[tid] = o_done ? 1 : 0;
      }
      __syncthreads();
      ESPM_PHASE_STAMP(2);   // numerators, denominators, bracket
      int evals = 0;
      // per-wave partial sums of f_kk and f_kk' at delta = s_x[kk] into bis[evals & 1]; ends with a barrier
      auto evaluate = [&]() {
        double* sc = bis[evals & 1];
        ++evals;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
          double f = 0.0, fp = 0.0;
          if (kk < k) {
            const double at = s_x[kk] - s_dstar[kk];   // (delta + den - d*: den - d* >= 0 is exact in fp64 for fp32 inputs of one scale)
#pragma unroll
            for (int r = 0; r < WF_ROWS; ++r) {
              if (tid + r * NT < M && in_set[r]) {
                const double inv = rcp_f64(at + (double)dv[r][kk]);
                const double t = nv[r][kk] > 0.f ? (double)nv[r][kk] * inv : 0.0;
                if (t > (double)a.log_shift) {
                  f += t;
                  fp -= t * inv;
                } else {
                  f += (double)a.log_shift;
                }
              }
            }
          }
          f = wave_sum(f);
          fp = wave_sum(fp);
          if (lane == 0) {
            sc[wave * 2 * KA + kk] = f;
            sc[wave * 2 * KA + KA + kk] = fp;
          }
        }
        __syncthreads();
        return sc;
      };
      auto combine = [&](const double* sc, int kk, double& fs, double& fps) {   // wave order: deterministic
        fs = sc[kk];
        fps = sc[KA + kk];
        for (int w = 1; w < NWV; ++w) {
          fs += sc[w * 2 * KA + kk];
          fps += sc[w * 2 * KA + KA + kk];
        }
        fs -= 1.0;
      };
      for (int it = 0; it < 100; ++it) {   // 1. the roots
        const double* sc = evaluate();
        if (owner && !o_done) {
          double fs, fps;
          combine(sc, tid, fs, fps);
          o_fder = fps;
          if (fabs(fs) <= 1e-11) {
            o_done = true;
          } else {
            if (fs > 0) o_lo = o_x; else o_hi = o_x;
            double dx = fps < 0 ? -fs / fps : 0.0;
            double xn = o_x + dx;
            if (!(fps < 0) || !(xn > o_lo && xn < o_hi) || fabs(dx) > 0.5 * fabs(o_dxold)) {
              dx = (o_hi - o_lo) / 2;
              xn = o_lo + dx;
            }
            o_dxold = dx;
            if (xn == o_x || fabs(dx) <= 1e-15 * fabs(o_x)) o_done = true;
            o_x = xn;
          }
          s_x[tid] = o_x;
          s_flag[tid] = o_done ? 1 : 0;
        }
        __syncthreads();
        bool all_done = true;
        for (int kk = 0; kk < k; ++kk) all_done = all_done && s_flag[kk] != 0;
        if (all_done) break;
      }
      // 2., 3. the reference's sweep count and its midpoints (in delta: a + d* + (b - a) frac)
      if (owner) {
        const double ad = s_lo[tid] + s_dstar[tid], width = s_hi[tid] - s_lo[tid];
        s_root[tid] = o_x;
        s_fder[tid] = solve ? o_fder : 0.0;
        s_ad[tid] = ad;
        s_width[tid] = width;
        s_u[tid] = solve ? fmin(fmax((o_x - ad) / width, 0.0), 1.0) : 0.5;
      }
      __syncthreads();
      // midpoint of sweep t (the first midpoint is t = 1) on the way to a root at fraction u of the bracket, as a fraction
      auto mid_frac = [](double uu, int t) {
        const double scale = ldexp(1.0, t - 1);
        const double cell = fmin(floor(uu * scale), scale - 1.0);
        return ldexp(2.0 * cell + 1.0, -t);
      };
      ESPM_PHASE_STAMP(3);   // roots found
#ifdef ESPM_PHASE_CLOCK
      if (threadIdx.x == 0 && espm_phase_buf) espm_phase_buf[8] = (unsigned long long)evals;
#endif
      // the sweeps are independent given the roots: lane l of wave 0 looks at sweeps l + 1 and l + 65 (dicotomy.py:152: at
      // most maxit = 100 sweeps after the first midpoint); two bit masks - "certainly stops here", "within 1 % of tol"
      if (wave == 0) {
        for (int hf = 0; hf < 2; ++hf) {
          const int t = 1 + lane + 64 * hf;
          double est = 0.0;
          for (int kk = 0; kk < k; ++kk)
            est = fmax(est, fabs(s_fder[kk] * ((s_ad[kk] + s_width[kk] * mid_frac(s_u[kk], t)) - s_root[kk])));
          const bool certain = t <= 101 && (est <= 0.99 * tol || t == 101);
          const bool band = t <= 101 && !certain && est <= 1.01 * tol;
          const unsigned long long mc = __ballot(certain), mb = __ballot(band);
          if (lane == 0) {
            s_mask[hf] = mc;
            s_mask[2 + hf] = mb;
          }
        }
      }
      __syncthreads();
      unsigned long long mc[2] = {s_mask[0], s_mask[1]}, mb[2] = {s_mask[2], s_mask[3]};
      int t_stop = 101;
      for (;;) {   // (uniform: every thread reads the same masks and the same sums)
        const unsigned long long w0 = mc[0] | mb[0], w1 = mc[1] | mb[1];
        if (!w0 && !w1) break;
        const int hf = w0 ? 0 : 1;
        const int bit = __ffsll((long long)(hf ? w1 : w0)) - 1;
        const int t = 1 + bit + 64 * hf;
        if ((mc[hf] >> bit) & 1ull) { t_stop = t; break; }
        if (owner) s_x[tid] = s_ad[tid] + s_width[tid] * mid_frac(s_u[tid], t);
        __syncthreads();
        const double* sc = evaluate();
        double worst = 0.0;
        for (int kk = 0; kk < k; ++kk) {
          double fs, fps;
          combine(sc, kk, fs, fps);
          if (s_fder[kk] != 0.0) worst = fmax(worst, fabs(fs));
        }
        if (worst <= tol) { t_stop = t; break; }
        mb[hf] &= ~(1ull << bit);
      }
      if (owner) s_mid[tid] = solve ? s_ad[tid] + s_width[tid] * mid_frac(s_u[tid], t_stop) : s_dstar[tid];   // delta = nu + d* (no multiplier: nu = 0)
      __syncthreads();
    }
    ESPM_PHASE_STAMP(4);   // the sweep the reference stops at
#ifdef ESPM_PHASE_CLOCK
    if (a.simplex_w && threadIdx.x == 0 && espm_phase_buf) espm_phase_buf[9] = 1000ull;   // (marks a simplex call)
#endif
    // W' = max(num / (den + nu), eps), fixed entries (updates.py:70-76); rel_W (base.py:323)
    double sum_l = 0.0;
#pragma unroll
    for (int r = 0; r < WF_ROWS; ++r) {
      const int mm = tid + r * NT;
      if (mm < M) {
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
          if (kk < k) {
            float den = dv[r][kk];
            if (a.simplex_w && in_set[r]) den = (float)(((double)den - s_dstar[kk]) + s_mid[kk]);
            float v = fmaxf(nv[r][kk] / den, a.log_shift);
            if (a.fixed_w) {
              const float fx = a.fixed_w[mm * k + kk];
              if (fx >= 0.f) v = fx;
            }
            wn[r][kk] = v;
            a.w_new[mm * k + kk] = v;
            if (a.g) s_w[mm * SW + kk] = v;
            sum_l += (double)v;
          }
        }
      }
    }
    const double mean_w = block_sum1(sum_l, scratch) / (double)MK;
    double rel_l = 0.0;
#pragma unroll
    for (int r = 0; r < WF_ROWS; ++r) {
      if (tid + r * NT < M) {
#pragma unroll
        for (int kk = 0; kk < KA; ++kk)
          if (kk < k)
            rel_l = fmax(rel_l, fabs((double)wn[r][kk] - (double)wo[r][kk]) / ((double)wn[r][kk] + (double)a.rel_tol * mean_w));
      }
    }
    const double rel_w = block_max1(rel_l, scratch);
    if (tid == 0 && a.hist_slot) a.hist_slot[ESPM_HI_REL_W] = rel_w;
    if (a.pg_q) {  // (uniform) the projected gradient's linesearch term sum <W' - W, grad> + gamma ||W' - W||^2
      double q_l = 0.0;
#pragma unroll
      for (int r = 0; r < WF_ROWS; ++r) {
        if (tid + r * NT < M) {
#pragma unroll
          for (int kk = 0; kk < KA; ++kk)
            if (kk < k) {
              const double dw = (double)wn[r][kk] - (double)wo[r][kk];
              q_l += dw * (double)pgrad[r][kk] + (double)a.pg_gamma_w * dw * dw;
            }
        }
      }
      const double q_w = block_sum1(q_l, scratch);
      if (tid == 0) *a.pg_q = q_w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < WF_ROWS; ++r) {
      const int mm = tid + r * NT;
      if (mm < M) {
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
          if (kk < k) {
            wn[r][kk] = a.w_new[mm * k + kk];
            if (a.g) s_w[mm * SW + kk] = wn[r][kk];
          }
        }
      }
    }
    __syncthreads();
  }

  ESPM_PHASE_STAMP(5);   // W', rel_W
  // GW = G W' (updates.py:107 of the next half step), stored / xscale with a positive floor
  double cs[KA];
#pragma unroll
  for (int kk = 0; kk < KA; ++kk) cs[kk] = 0.0;
  const float inv_scale = 1.f / a.xscale;
  auto emit_row = [&](int c, const float (&src)[KA]) {
    float row[espm::KP];
#pragma unroll
    for (int kk = 0; kk < espm::KP; ++kk) row[kk] = 0.f;
    if (c >= a.n) {
#pragma unroll
      for (int kk = 0; kk < KA; ++kk) row[kk] = 1.f;  // padding channels: X = 0 there
    } else {
#pragma unroll
      for (int kk = 0; kk < KA; ++kk) {
        const float v = fmaxf(src[kk], a.gw_floor);
        cs[kk] += (double)v;
        row[kk] = v * inv_scale;
      }
    }
    if (c < a.n_pad) {
      espm::store_row_kp(a.gw_s + (size_t)c * espm::KP, row);
    }
  };
  const int n_rows = a.n_pad;
  if (a.g) {
    for (int c = tid; c < n_rows; c += NT) {
      float row[KA];
#pragma unroll
      for (int kk = 0; kk < KA; ++kk) row[kk] = 0.f;
      if (c < a.n) {
        constexpr int MB = 8;  // entries of the G row requested together
        for (int m0 = 0; m0 < a.m; m0 += MB) {
          float gv[MB];
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            const int mm = m0 + b < a.m ? m0 + b : a.m - 1;
            gv[b] = a.g_t ? a.g_t[(size_t)mm * a.n_pad + c] : a.g[(size_t)c * a.m + mm];
          }
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            if (m0 + b < a.m) {
              // a row of W' as aligned 16-byte reads (one 4-byte read per multiply-add was 16 us of this kernel's 36 at C5)
              const float4* wr = reinterpret_cast<const float4*>(s_w + (size_t)(m0 + b) * SW);
#pragma unroll
              for (int q4 = 0; q4 < SW / 4; ++q4) {
                const float4 w4 = wr[q4];
                const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (4 * q4 + i < KA) row[4 * q4 + i] = fmaf(gv[b], wv[i], row[4 * q4 + i]);
              }
            }
          }
        }
      }
      emit_row(c, row);
    }
  } else {  // G = identity: row c of G W' is row c of W', already in this thread's registers
#pragma unroll
    for (int r = 0; r < WF_ROWS; ++r) {
      const int c = tid + r * NT;
      if (c < n_rows) emit_row(c, wn[r]);
    }
  }
  block_reduce<KA, KA>(cs, scratch);
  if (tid == 0)
    for (int kk = 0; kk < espm::KP; ++kk) a.colsum_gw[kk] = kk < KA ? cs[kk] : 0.0;
  ESPM_PHASE_STAMP(6);   // rows of G W', column sums
#ifdef ESPM_PHASE_CLOCK
  if (threadIdx.x == 0 && espm_phase_buf) espm_phase_buf[21] = (unsigned long long)clock64();
#endif
}

__global__ __launch_bounds__(WF_THREADS) void w_finish_kernel(const WFinishArgs a) {
  __shared__ double scratch[(WF_THREADS / 64 + 1) * 2 * KP];
  __shared__ double s_lo[KP], s_hi[KP], s_mid[KP], s_f[KP];
  __shared__ int s_go;
  const int M = a.m > 0 ? a.m : a.n;
  const int k = a.k;
  const int tid = threadIdx.x;
  float* numv = a.scratch;
  float* denv = a.scratch + (size_t)M * k;

  if (a.update_w) {
    // numerator W * (G^T A) and denominator colsum(G) rowsum(H)^T, updates.py:58-60
    for (int e = tid; e < M * k; e += WF_THREADS) {
      const int mm = e / k, kk = e - mm * k;
      float gta;
      if (a.g) {
        float s = 0.f;
        for (int c = 0; c < a.n; ++c) s = fmaf(a.g[(size_t)c * a.m + mm], load_a(a, kk, c), s);
        gta = s;
      } else {
        gta = load_a(a, kk, mm);
      }
      const float wo = a.w_old[e];
      float nvv = wo * gta;
      float dvv = (a.g ? a.colsum_g[mm] : 1.f) * (float)a.hstat[ESPM_HS_ROWSUM + kk];
      if (a.pg_gamma_w > 0.f) {           // projected gradient: W - (colsum(G) rowsum(H) - G^T A) / gamma, updates.py:353-362 (as w_finish_fast_kernel)
        const float pgr = dvv - gta;
        nvv = wo - pgr / a.pg_gamma_w;
        dvv = pgr;                        // (the denominator is 1: its slot carries the gradient for the linesearch term below)
      } else if (a.breg_sr) {             // Bregman variant (G = identity), updates.py:41-48
        const float sr = a.xscale * a.breg_sr[mm];
        dvv = (dvv - gta) * wo + sr;
        nvv = sr * wo;
      }
      numv[e] = nvv;
      denv[e] = dvv;
    }
    __syncthreads();

    if (a.simplex_w) {
      // bracket of dicotomy.py:29-49 per column kk over the constrained rows
      double cnt_l = 0.0;
      for (int mm = tid; mm < M; mm += WF_THREADS) cnt_l += (!a.simplex_rows || a.simplex_rows[mm]) ? 1.0 : 0.0;
      const double rows = block_sum1(cnt_l, scratch);
      for (int kk = 0; kk < k; ++kk) {
        double lo = -INFINITY, nmax = 0.0, dmin_neg = -INFINITY;
        for (int mm = tid; mm < M; mm += WF_THREADS) {
          if (a.simplex_rows && !a.simplex_rows[mm]) continue;
          const double nn = numv[mm * k + kk], dd = denv[mm * k + kk];
          if (nn > 0) lo = fmax(lo, nn / 2 - dd);
          nmax = fmax(nmax, nn);
          dmin_neg = fmax(dmin_neg, -dd);
        }
        lo = block_max1(lo, scratch);
        nmax = block_max1(nmax, scratch);
        dmin_neg = block_max1(dmin_neg, scratch);
        if (tid == 0) {
          s_lo[kk] = lo;
          s_hi[kk] = rows * nmax / 0.5 + dmin_neg;
        }
      }
      __syncthreads();
      // bisection with the reference's global stop rule, dicotomy.py:146-171
      for (int it = 0; it <= 100; ++it) {
        if (tid < k) s_mid[tid] = (s_lo[tid] + s_hi[tid]) / 2;
        __syncthreads();
        double f[KP];
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) f[kk] = 0.0;
        for (int mm = tid; mm < M; mm += WF_THREADS) {
          if (a.simplex_rows && !a.simplex_rows[mm]) continue;
#pragma unroll
          for (int kk = 0; kk < KP; ++kk)
            if (kk < k)
              f[kk] += fmax((double)numv[mm * k + kk] / (s_mid[kk] + (double)denv[mm * k + kk]), (double)a.log_shift);
        }
        block_reduce<KP, KP>(f, scratch);
        if (tid == 0) {
          double worst = 0.0;
          for (int kk = 0; kk < k; ++kk) {
            s_f[kk] = f[kk] - 1.0;
            worst = fmax(worst, fabs(s_f[kk]));
          }
          s_go = (worst > (double)a.tol) && (it < 100);
          if (s_go) {
            for (int kk = 0; kk < k; ++kk) {
              if (s_f[kk] <= 0.0) s_hi[kk] = s_mid[kk]; else s_lo[kk] = s_mid[kk];
            }
          }
        }
        __syncthreads();
        if (!s_go) break;
      }
    }

    // W' = max(num / (den + nu), eps), fixed entries, updates.py:70-76
    double sum_l = 0.0, q_l = 0.0;
    const bool pg = a.pg_gamma_w > 0.f;   // (never with the simplex over W: espm_mu's state check)
    for (int e = tid; e < M * k; e += WF_THREADS) {
      const int mm = e / k, kk = e - mm * k;
      float den = pg ? 1.f : denv[e];
      if (a.simplex_w && (!a.simplex_rows || a.simplex_rows[mm])) den += (float)s_mid[kk];
      float wn = fmaxf(numv[e] / den, a.log_shift);
      if (a.fixed_w && a.fixed_w[e] >= 0.f) wn = a.fixed_w[e];
      a.w_new[e] = wn;
      sum_l += (double)wn;
      if (pg) {   // the linesearch term sum <W' - W, grad> + gamma ||W' - W||^2
        const double dw = (double)wn - (double)a.w_old[e];
        q_l += dw * (double)denv[e] + (double)a.pg_gamma_w * dw * dw;
      }
    }
    if (a.pg_q) {   // (uniform)
      const double q_w = block_sum1(q_l, scratch);
      if (tid == 0) *a.pg_q = q_w;
    }
    const double mean_w = block_sum1(sum_l, scratch) / ((double)M * k);
    double rel_l = 0.0;
    for (int e = tid; e < M * k; e += WF_THREADS) {
      const double wn = a.w_new[e], wo = a.w_old[e];
      rel_l = fmax(rel_l, fabs(wn - wo) / (wn + (double)a.rel_tol * mean_w));  // base.py:323
    }
    const double rel_w = block_max1(rel_l, scratch);
    if (tid == 0 && a.hist_slot) a.hist_slot[ESPM_HI_REL_W] = rel_w;
  }

  // GW = G W' (updates.py:107 of the next half step), stored / xscale with a positive floor
  const float* w = a.w_new;
  double cs[KP];
#pragma unroll
  for (int kk = 0; kk < KP; ++kk) cs[kk] = 0.0;
  const float inv_scale = 1.f / a.xscale;
  for (int c = tid; c < a.n_pad; c += WF_THREADS) {
    float row[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      float v = 0.f;
      if (kk < k) {
        if (c >= a.n) {
          v = 1.f;  // padding channels: X = 0 there, any positive value keeps X / Y = 0
        } else {
          if (a.g) {
            for (int mm = 0; mm < a.m; ++mm) v = fmaf(a.g[(size_t)c * a.m + mm], w[mm * k + kk], v);
          } else {
            v = w[c * k + kk];
          }
          v = fmaxf(v, a.gw_floor);
          cs[kk] += (double)v;
          v *= inv_scale;
        }
      }
      row[kk] = v;
    }
    store_row_kp(a.gw_s + (size_t)c * KP, row);
  }
  block_reduce<KP, KP>(cs, scratch);
  if (tid == 0)
    for (int kk = 0; kk < KP; ++kk) a.colsum_gw[kk] = cs[kk];
}

// ---- dispatch -----------------------------------------------------------------------------------
template <int K>
static int dispatch_w_k(const WAccumArgs& args, int x_dtype, int nblk, hipStream_t stream) {
  if (args.l2 && x_dtype != ESPM_X_F32) return set_error(ESPM_EUNSUPPORTED, "the l2 W accumulation needs the f32 store");
  // channels per lane: 8 up to 8 components; 4 beyond (8 x k accumulators and 8 x k GW entries do not fit the registers)
  constexpr int CH8 = K <= 8 ? 8 : 4;
  // both contractions on the matrix cores (mu_w_mfma_kernel.hpp) from MFMA_MIN_K components on: below, the vector kernel's 2 k + 6
  // instructions per element are fewer than what the split operands and the tile traffic cost
  if (K >= ESPM_MFMA_MIN_K && args.mfma && args.x_cm && args.n_pad % 8 == 0) {
    const dim3 grid(nblk, (args.n_pad + 4 * 16 * MF_CT - 1) / (4 * 16 * MF_CT));
    if (x_dtype == ESPM_X_U8) {
      hipLaunchKernelGGL((w_accum_mfma_kernel<K, uint8_t>), grid, dim3(256), 0, stream, args);
    } else if (x_dtype == ESPM_X_BF16) {
      hipLaunchKernelGGL((w_accum_mfma_kernel<K, bf16_t>), grid, dim3(256), 0, stream, args);
    } else if (args.l2) {
      hipLaunchKernelGGL((w_accum_mfma_l2_kernel<K>), grid, dim3(256), 0, stream, args);
    } else {
      hipLaunchKernelGGL((w_accum_mfma_kernel<K, float>), grid, dim3(256), 0, stream, args);
    }
    return check_hip(hipGetLastError(), "w_accum (mfma) launch");
  }
  if constexpr (K > 16) {   // the widest build keeps the vector kernel for the fp32 store only (mu_h_step.hip)
    if (x_dtype != ESPM_X_F32)
      return set_error(ESPM_EUNSUPPORTED, "w_accum: %d components on the 8-bit / bf16 store run on the matrix cores only (no_fused = 0, n_pad a multiple of 8)", K);
  }
  if (x_dtype == ESPM_X_U8) {
    if constexpr (K <= 16) {
      dim3 grid(nblk, (args.n_pad + 4 * 64 * CH8 - 1) / (4 * 64 * CH8));
      hipLaunchKernelGGL((w_accum_kernel<K, uint8_t, CH8, 4, 3>), grid, dim3(256), 0, stream, args);  // ring of 3: tools/tune
    }
  } else if (x_dtype == ESPM_X_BF16) {
    if constexpr (K <= 16) {
      dim3 grid(nblk, (args.n_pad + 4 * 64 * CH8 - 1) / (4 * 64 * CH8));
      hipLaunchKernelGGL((w_accum_kernel<K, bf16_t, CH8, 4, 0>), grid, dim3(256), 0, stream, args);
    }
  } else {
    dim3 grid(nblk, (args.n_pad + 4 * 64 * 4 - 1) / (4 * 64 * 4));
    if (args.l2)
      hipLaunchKernelGGL((w_accum_kernel<K, float, 4, 4, 0, true>), grid, dim3(256), 0, stream, args);
    else
      hipLaunchKernelGGL((w_accum_kernel<K, float, 4, 4, 0>), grid, dim3(256), 0, stream, args);
  }
  return check_hip(hipGetLastError(), "w_accum launch");
}

int dispatch_w_accum(const WAccumArgs& args, int k, int x_dtype, int nblk, hipStream_t stream) {
  switch (k) {
#define ESPM_X(KK) case KK: return dispatch_w_k<KK>(args, x_dtype, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
  }
  return set_error(ESPM_EUNSUPPORTED, "w_accum: k=%d not built (%d..%d)", k, ESPM_MIN_K, ESPM_MAX_K);
}

int launch_w_reduce(const float* slab, float* out, int nblk, int total, const HFinalizeArgs* fused_finalize,
                    hipStream_t stream, const float* bw_old, double* bparts, int n, int k, int n_pad) {
  WReduceArgs a;
  a.bw_old = bw_old;
  a.bparts = bparts;
  a.bn = n;
  a.bk = k;
  a.bn_pad = n_pad;
  a.slab = slab;
  a.out = out;
  a.nblk = nblk;
  a.total = total;
  a.nred_blocks = (total + 31) / 32;
  a.fuse_finalize = fused_finalize != nullptr;
  if (fused_finalize) a.fin = *fused_finalize;
  a.halo_h = nullptr;
  a.halo_top = a.halo_bot = nullptr;
  a.halo_k = a.halo_nx = a.halo_ny = a.halo_ppad = 0;
  hipLaunchKernelGGL(w_reduce_kernel, dim3(a.nred_blocks + (fused_finalize ? H_FINALIZE_JOBS : 0)), dim3(256), 0, stream, a);
  return check_hip(hipGetLastError(), "w_reduce launch");
}

// Slab reduction straight into a rank's record: A block, statistics of the new H (the finalize workgroup writes them
// there), boundary rows of the new H.
int launch_w_reduce_pack(const float* slab, int nblk, int k, int n_pad, const HFinalizeArgs& fin_to_record,
                         const float* h_new, int nx, int ny, int p_pad, int with_halo, void* rec, hipStream_t stream) {
  WReduceArgs a;
  unsigned char* r = static_cast<unsigned char*>(rec);
  const int na = k * n_pad;
  a.slab = slab;
  a.out = reinterpret_cast<float*>(r);
  a.nblk = nblk;
  a.total = na;
  a.nred_blocks = (na + 31) / 32;
  a.fuse_finalize = 1;
  a.fin = fin_to_record;
  a.halo_h = h_new;
  a.halo_top = with_halo ? reinterpret_cast<float*>(r + (size_t)na * 4 + ESPM_HS_STRIDE * 8) : nullptr;
  a.halo_bot = with_halo ? a.halo_top + (size_t)k * ny : nullptr;
  a.halo_k = k;
  a.halo_nx = nx;
  a.halo_ny = ny;
  a.halo_ppad = p_pad;
  a.bw_old = nullptr;
  a.bparts = nullptr;
  a.bn = a.bk = a.bn_pad = 0;
  hipLaunchKernelGGL(w_reduce_kernel, dim3(a.nred_blocks + H_FINALIZE_JOBS), dim3(256), 0, stream, a);
  return check_hip(hipGetLastError(), "w_reduce_pack launch");
}

// the tail's view of a local W update: partials in f.scratch, W before / after, where the results go
WTailArgs make_w_tail_args(const WFinishArgs& f) {
  WTailArgs t;
  t.parts = reinterpret_cast<const double*>(f.scratch);
  t.w_old = f.w_old;
  t.w_new = f.w_new;
  t.colsum_gw = f.colsum_gw;
  t.hist_slot = f.hist_slot;
  t.pg_q = f.pg_q;
  t.n = f.n;
  t.k = f.k;
  t.nbk = (f.n_pad + 31) / 32;
  t.rel_tol = f.rel_tol;
  return t;
}

int launch_w_reduce_update(const WFinishArgs& f, const void* src, size_t src_stride, int nsrc, float* a_out,
                           const double* hpart, int nblk_h, const double* hstat_rs, size_t rec_hstat_off, double* hstat_out,
                           const HFinalizeArgs* fused_finalize, hipStream_t stream, WTailArgs* defer_tail) {
  WUpdateArgs a;
  a.src = static_cast<const unsigned char*>(src);
  a.src_stride = src_stride;
  a.nsrc = nsrc;
  a.n = f.n;
  a.n_pad = f.n_pad;
  a.k = f.k;
  a.nbk = (f.n_pad + 31) / 32;
  a.a_out = a_out;
  a.hpart = hpart;
  a.hstat_rs = hstat_rs;
  a.nblk_h = nblk_h;
  a.rec_hstat_off = rec_hstat_off;
  a.hstat_out = hstat_out;
  a.w_old = f.w_old;
  a.w_new = f.w_new;
  a.fixed_w = f.fixed_w;
  a.breg_sr = f.breg_sr;
  a.pg_gamma_w = f.pg_gamma_w;
  a.pg_track = f.pg_q != nullptr;
  a.gw_s = f.gw_s;
  a.parts = reinterpret_cast<double*>(f.scratch);
  a.log_shift = f.log_shift;
  a.gw_floor = f.gw_floor;
  a.xscale = f.xscale;
  a.fuse_finalize = fused_finalize != nullptr;
  if (fused_finalize) a.fin = *fused_finalize;
  hipLaunchKernelGGL(w_reduce_update_kernel, dim3(a.k * a.nbk + (fused_finalize ? H_FINALIZE_JOBS : 0)), dim3(256), 0, stream, a);
  const WTailArgs t = make_w_tail_args(f);
  if (defer_tail)   // (espm_mu_iterate: the tail rides in the next H-step's launch, or in launch_w_update_tail at the end)
    *defer_tail = t;
  else
    hipLaunchKernelGGL(w_update_tail_kernel, dim3(1), dim3(WT_THREADS), 0, stream, t);
  return check_hip(hipGetLastError(), "w_reduce_update launch");
}

int launch_w_simplex_update(const WFinishArgs& f, float* a_inout, const double* bparts, double tol, hipStream_t stream, WTailArgs* defer_tail) {
  WSimplexArgs x;
  WUpdateArgs& a = x.u;
  a.src = nullptr;
  a.src_stride = 0;
  a.nsrc = 0;
  a.n = f.n;
  a.n_pad = f.n_pad;
  a.k = f.k;
  a.nbk = (f.n_pad + 31) / 32;
  a.a_out = a_inout;
  a.hpart = nullptr;
  a.hstat_rs = nullptr;
  a.nblk_h = 0;
  a.rec_hstat_off = 0;
  a.hstat_out = nullptr;
  a.w_old = f.w_old;
  a.w_new = f.w_new;
  a.fixed_w = f.fixed_w;
  a.breg_sr = nullptr;
  a.pg_gamma_w = 0.f;
  a.pg_track = 0;
  a.gw_s = f.gw_s;
  a.parts = reinterpret_cast<double*>(f.scratch);
  a.log_shift = f.log_shift;
  a.gw_floor = f.gw_floor;
  a.xscale = f.xscale;
  a.fuse_finalize = 0;
  x.bparts = bparts;
  x.hstat = f.hstat;
  x.rows = (double)f.n;
  x.tol = tol;
  hipLaunchKernelGGL(w_simplex_update_kernel, dim3(a.k * a.nbk), dim3(64), 0, stream, x);
  if (int rc = check_hip(hipGetLastError(), "w_simplex_update launch")) return rc;
  const WTailArgs t = make_w_tail_args(f);
  if (defer_tail)
    *defer_tail = t;
  else
    hipLaunchKernelGGL(w_update_tail_kernel, dim3(1), dim3(WT_THREADS), 0, stream, t);
  return check_hip(hipGetLastError(), "w_simplex_update tail launch");
}

int launch_w_exchange_update(const WFinishArgs& f, const void* slabs, size_t slab_stride, int nslab, float* a_out, double* hstat_out,
                             const HFinalizeArgs& fin, const espm_xchg* xc, unsigned int seq, const float* h_new, int nx, int ny, int p_pad,
                             int with_halo, hipStream_t stream, WTailArgs* defer_tail, double* bparts) {
  WExchangeArgs x;
  x.bparts = bparts;
  WUpdateArgs& a = x.u;
  a.src = static_cast<const unsigned char*>(slabs);
  a.src_stride = slab_stride;
  a.nsrc = nslab;
  a.n = f.n;
  a.n_pad = f.n_pad;
  a.k = f.k;
  a.nbk = (f.n_pad + 31) / 32;
  a.a_out = a_out;
  a.hpart = fin.hpart;       // (every reduction workgroup forms the row sum it needs from the H-step's records itself)
  a.hstat_rs = nullptr;
  a.nblk_h = fin.nblk;
  a.rec_hstat_off = 0;
  a.hstat_out = hstat_out;
  a.w_old = f.w_old;
  a.w_new = f.w_new;
  a.fixed_w = f.fixed_w;
  a.breg_sr = f.breg_sr;
  a.pg_gamma_w = f.pg_gamma_w;
  a.pg_track = f.pg_q != nullptr;
  a.gw_s = f.gw_s;
  a.parts = reinterpret_cast<double*>(f.scratch);
  a.log_shift = f.log_shift;
  a.gw_floor = f.gw_floor;
  a.xscale = f.xscale;
  a.fuse_finalize = 1;
  a.fin = fin;
  const int nwg = a.k * a.nbk;
  ESPM_REQUIRE(nwg + 1 <= xc->wgflags, "exchange: %d reduction workgroups, the mailbox holds flags for %d", nwg, xc->wgflags - 1);
  for (int r = 0; r < 16; ++r) x.mbox[r] = r < xc->world ? xc->peers[r] : nullptr;
  for (int r = 0; r < xc->world; ++r) ESPM_REQUIRE(x.mbox[r], "exchange: rank %d is not connected (espm_xchg_connect)", r);
  x.world = xc->world;
  x.rank = xc->rank;
  x.nfl = xc->wgflags;
  x.with_halo = with_halo;
  x.rec_bytes = xc->record_bytes;
  x.slot_base = (size_t)(seq & 1u) * xc->world * xc->record_bytes;
  x.wgflags_off = xc->off_wgflags;
  x.gran_off = xc->off_gran + (size_t)(seq & 1u) * xc->world * ((size_t)34 * xc->wgflags + 2 * ESPM_HS_STRIDE) * sizeof(unsigned long long);
  x.err_off = xc->off_err;
  x.hstat_off = (size_t)f.k * f.n_pad * 4;
  x.top_off = x.hstat_off + ESPM_HS_STRIDE * 8;
  x.bot_off = x.top_off + (size_t)f.k * (ny > 0 ? ny : 0) * 4;
  x.seq = seq;
  x.max_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  x.release = xc->order;
  x.halo_h = h_new;
  x.halo_k = f.k;
  x.halo_nx = nx;
  x.halo_ny = ny;
  x.halo_ppad = p_pad;
  a.fin.hstat_out = reinterpret_cast<double*>(xc->mailbox + x.slot_base + (size_t)xc->rank * xc->record_bytes + x.hstat_off);
  if (xc->world <= ESPM_XCHG_SMALL_WORLD)
    hipLaunchKernelGGL(w_exchange_update_kernel<ESPM_XCHG_SMALL_WORLD>, dim3(nwg + H_FINALIZE_JOBS), dim3(256), 0, stream, x);
  else
    hipLaunchKernelGGL(w_exchange_update_kernel<16>, dim3(nwg + H_FINALIZE_JOBS), dim3(256), 0, stream, x);
  if (bparts) return check_hip(hipGetLastError(), "w_exchange_update launch");   // (the caller's w_simplex_update_kernel updates W and owns the tail)
  const WTailArgs t = make_w_tail_args(f);
  if (defer_tail)
    *defer_tail = t;
  else
    hipLaunchKernelGGL(w_update_tail_kernel, dim3(1), dim3(WT_THREADS), 0, stream, t);
  return check_hip(hipGetLastError(), "w_exchange_update launch");
}

int launch_w_update_tail(const WTailArgs& t, hipStream_t stream) {
  hipLaunchKernelGGL(w_update_tail_kernel, dim3(1), dim3(WT_THREADS), 0, stream, t);
  return check_hip(hipGetLastError(), "w_update_tail launch");
}

template <int KK, int NT>
static void launch_fast(const WFinishArgs& args, int rows, int crows, size_t lds, hipStream_t stream) {
  if (crows > rows && rows <= 1) {   // dictionary G with at most NT rows: W state of one row per thread, 2 or 4 channels per thread for G^T A
    if (crows <= 2)
      hipLaunchKernelGGL((w_finish_fast_kernel<KK, 1, NT, 2>), dim3(1), dim3(NT), lds, stream, args);
    else
      hipLaunchKernelGGL((w_finish_fast_kernel<KK, 1, NT, 4>), dim3(1), dim3(NT), lds, stream, args);
    return;
  }
  if (crows > rows) rows = crows;
  if constexpr (NT == 256) {   // (4 waves: 8 rows per thread at 2048 channels)
    if (rows > 4) {
      hipLaunchKernelGGL((w_finish_fast_kernel<KK, 8, NT>), dim3(1), dim3(NT), lds, stream, args);
      return;
    }
  }
  if (rows <= 1)
    hipLaunchKernelGGL((w_finish_fast_kernel<KK, 1, NT>), dim3(1), dim3(NT), lds, stream, args);
  else if (rows <= 2)
    hipLaunchKernelGGL((w_finish_fast_kernel<KK, 2, NT>), dim3(1), dim3(NT), lds, stream, args);
  else
    hipLaunchKernelGGL((w_finish_fast_kernel<KK, 4, NT>), dim3(1), dim3(NT), lds, stream, args);
}

// ---- W finish with a dictionary G (n x m, few columns) as TWO launches (C5: 1980 x 17, k = 8) ----------------------------------
// The one-workgroup finish spends its 24-36 us on one CU: the m k dot products G^T A over the n channels (butterfly sums across a
// wave) and the rows of G W'.  The W update needs nothing global but the row sums of H' (no simplex over W here), so:
//   w_gfinish_update_kernel  one workgroup per entry (row of G, component): G^T A over the channels from the transposed copy of G
//                            (coalesced), a fixed-order block sum, W' = max(W (G^T A) / (colsum(G) rowsum(H')), eps), fixed_W
//                            (updates.py:58-60, :70-76)
//   w_gfinish_gw_kernel      256 channels per workgroup: W' (m k floats) into LDS, the rows of G W' (a thread per channel: m k
//                            multiply-adds from LDS, its row of G requested with W'); workgroup 0 forms the column sums of G W' as
//                            colsum(G)^T W' (round 4: per-workgroup partials + a ticket before), mean(W') and rel_W (base.py:323)
// ESPM_W_GSPLIT=0 keeps the one-workgroup finish (A/B).
__global__ __launch_bounds__(256) void w_gfinish_update_kernel(const WFinishArgs a) {
  __shared__ float s_wave[4];
  const int e = blockIdx.x, mm = e / a.k, kk = e - mm * a.k;
  const float* gt = a.g_t + (size_t)mm * a.n_pad;
  const float* av = a.a + (size_t)kk * a.n_pad;
  float s = 0.f;
  for (int c0 = threadIdx.x; c0 < a.n; c0 += 8 * 256) {   // eight channels per thread requested together (a loop of single loads waits for each)
    float gv[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * 256;
      gv[u] = c < a.n ? gt[c] : 0.f;
      vv[u] = c < a.n ? av[c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fmaf(gv[u], vv[u], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gta = ((s_wave[0] + s_wave[1]) + s_wave[2]) + s_wave[3];
    const float num = a.w_old[e] * gta;
    const float den = a.colsum_g[mm] * (float)a.hstat[ESPM_HS_ROWSUM + kk];
    float wn = fmaxf(num / den, a.log_shift);
    if (a.fixed_w && a.fixed_w[e] >= 0.f) wn = a.fixed_w[e];
    a.w_new[e] = wn;
  }
}

// workgroup b: channels 256 b .. 256 b + 255, one per thread; partial column sums -> scratch; the workgroup that finishes last adds
// them in workgroup order (the same bits whoever is last) and forms mean(W') and rel_W
__global__ __launch_bounds__(256) void w_gfinish_gw_kernel(const WFinishArgs a) {
  extern __shared__ float s_w[];   // [m * k] the new W
  __shared__ double scratch[(256 / 64 + 1) * 2 * KP];
  const int tid = threadIdx.x, mk = a.m * a.k, k = a.k;
  // Round 4: ONE trip to memory.  The rows of G do not depend on W': they are requested together with it.  And the column sums of
  // G W' - the denominator of the next H update, updates.py:132 - are sum_m colsum(G)[m] W'[m, k] (fp64, workgroup 0, from the copy of
  // W' it holds anyway) instead of the sum of the rows over the channels, which took per-workgroup partials, a drain, a ticket and
  // the last workgroup's agent-scope loads: four more dependent round trips in a launch that is nothing but latency (14.3 us for 8
  // workgroups at BASELINE configuration 5, profiles/r03l_ks_c5_128rows_kernel_stats.csv).  The two sums agree to the rounding of
  // the fp32 rows (~1e-7 relative; an entry below gw_floor counts as itself, not as the floor: < 1e-30 each).
  // The LAST workgroup has no rows: it forms the column sums, mean(W') and rel_W - two block reductions behind loads of their own (W',
  // W, colsum(G), all requested at its start) - next to the row workgroups instead of behind workgroup 0's rows.
  if (blockIdx.x == gridDim.x - 1) {
    float wo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wo[u] = (tid + 256 * u < mk && a.hist_slot) ? a.w_old[tid + 256 * u] : 1.f;   // (up to 1024 entries requested with W'; beyond that in the loop)
    for (int e = tid; e < mk; e += 256) s_w[e] = a.w_new[e];
    __syncthreads();
    if (tid < KP) {
      double v = 0.0;
      if (tid < k)
        for (int mm = 0; mm < a.m; ++mm) v += (double)a.colsum_g[mm] * (double)s_w[mm * k + tid];
      a.colsum_gw[tid] = v;
    }
    if (a.hist_slot) {   // mean(W') and rel_W (base.py:323) over the m k entries
      double sum_l = 0.0;
      for (int e = tid; e < mk; e += 256) sum_l += (double)s_w[e];
      const double mean_w = block_sum1(sum_l, scratch) / (double)mk;
      double rel_l = 0.0;
      for (int e = tid, u = 0; e < mk; e += 256, ++u) {
        const double wn = s_w[e], wov = u < 4 ? wo[u] : a.w_old[e];
        rel_l = fmax(rel_l, fabs(wn - wov) / (wn + (double)a.rel_tol * mean_w));
      }
      const double rel_w = block_max1(rel_l, scratch);
      if (tid == 0) a.hist_slot[ESPM_HI_REL_W] = rel_w;
    }
    return;
  }
  const int c = blockIdx.x * 256 + tid;
  constexpr int GM = 32;   // entries of a row of G requested together (a dictionary has a few tens of columns)
  float gv[GM];
  const float* gc = a.g_t + (c < a.n ? c : 0);   // (the transposed copy: the lanes of a wave read neighbouring channels of one column of G)
#pragma unroll
  for (int u = 0; u < GM; ++u) gv[u] = (c < a.n && u < a.m) ? gc[(size_t)u * a.n_pad] : 0.f;
  for (int e = tid; e < mk; e += 256) s_w[e] = a.w_new[e];
  __syncthreads();
  // GW = G W' (updates.py:107 of the next half step), stored / xscale with a positive floor
  const float inv_scale = 1.f / a.xscale;
  if (c < a.n_pad) {
    float row[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) row[kk] = (kk < k && c >= a.n) ? 1.f : 0.f;   // padding channels: X = 0 there, any positive value keeps X / Y = 0
    if (c < a.n) {
      for (int m0 = 0; m0 < a.m; m0 += GM) {
        if (m0 > 0) {   // (more than GM columns: the further ones in batches of their own)
#pragma unroll
          for (int u = 0; u < GM; ++u) gv[u] = m0 + u < a.m ? gc[(size_t)(m0 + u) * a.n_pad] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < GM; ++u)
          if (m0 + u < a.m) {
#pragma unroll
            for (int kk = 0; kk < KP; ++kk)
              if (kk < k) row[kk] = fmaf(gv[u], s_w[(m0 + u) * k + kk], row[kk]);
          }
      }
#pragma unroll
      for (int kk = 0; kk < KP; ++kk)
        if (kk < k) row[kk] = fmaxf(row[kk], a.gw_floor) * inv_scale;
    }
    store_row_kp(a.gw_s + (size_t)c * KP, row);
  }
}

// ---- the same on a SHARDED image: what crosses the links is G^T A (m k values), not A (k n) -------------------------------------
// G^T (sum over the ranks of A_r) = sum over the ranks of G^T A_r: every rank contracts its OWN A with G (the update kernel's dot
// products), sends the m k results - one granule each, with the statistics and the boundary rows as in w_exchange_update_kernel -
// and adds the ranks' values in rank order: the same W' on every rank.  For C5 a rank ships 136 granules instead of a 63 KB record,
// and the sharded W step is three launches behind the accumulation (slab reduction + record reduction, this kernel, the rows of
// G W') instead of seven (reduce + pack, post, wait, combine, and the finish).
// grid: workgroup 0 = the extra workgroup (boundary rows, statistics, their flag), workgroup 1 + e = entry e = (row of G, component).
struct WGxchgArgs {
  WFinishArgs f;
  const double* hstat_local;   // this rank's statistics of the new H (the slab reduction's finalize left them in its own record)
  double* hstat_out;           // the global ones
  unsigned char* mbox[16];
  int world, rank, nfl, with_halo;
  size_t rec_bytes, slot_base, wgflags_off, gran_off, err_off, top_off, bot_off;
  unsigned int seq;
  long long max_ticks;
  int release;
  const float* halo_h;
  int halo_k, halo_nx, halo_ny, halo_ppad;
};

__global__ __launch_bounds__(256) void w_gxchg_update_kernel(const WGxchgArgs x) {
  const WFinishArgs& a = x.f;
  __shared__ float s_wave[4];
  const int mk = a.m * a.k;
  auto record = [&](int dst, int src_rank) { return x.mbox[dst] + x.slot_base + (size_t)src_rank * x.rec_bytes; };
  auto flag = [&](int dst, int src_rank) {   // (the extra workgroup's flag: index nfl - 1 of a rank's flags)
    return reinterpret_cast<unsigned int*>(x.mbox[dst] + x.wgflags_off + ((size_t)src_rank * x.nfl + (x.nfl - 1)) * sizeof(unsigned int));
  };
  const size_t GRAN = (size_t)34 * x.nfl + 2 * ESPM_HS_STRIDE;
  auto gran = [&](int dst, int src_rank, int g) {
    return reinterpret_cast<unsigned long long*>(x.mbox[dst] + x.gran_off) + (size_t)src_rank * GRAN + g;
  };
  unsigned int* err = reinterpret_cast<unsigned int*>(x.mbox[x.rank] + x.err_off);
  if (blockIdx.x == 0) {   // boundary rows (16-byte write-through stores), statistics as granules, the rows' flag
    if (x.with_halo) {
      for (int d = -1; d <= 1; ++d) {
        const int r = x.rank + d;
        if (r < 0 || r >= x.world) continue;
        float* top = reinterpret_cast<float*>(record(r, x.rank) + x.top_off);
        float* bot = reinterpret_cast<float*>(record(r, x.rank) + x.bot_off);
        if ((x.halo_ny & 3) == 0) {
          typedef float xf4 __attribute__((ext_vector_type(4)));
          for (int e = 4 * threadIdx.x; e < x.halo_k * x.halo_ny; e += 4 * 256) {
            const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
            const xf4 vt = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + j);
            const xf4 vb = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(top + e), "v"(vt) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(bot + e), "v"(vb) : "memory");
          }
        } else {
          for (int e = threadIdx.x; e < x.halo_k * x.halo_ny; e += 256) {
            const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
            __hip_atomic_store(top + e, x.halo_h[(size_t)kk * x.halo_ppad + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(bot + e, x.halo_h[(size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
    for (int i = threadIdx.x; i < x.world * 2 * ESPM_HS_STRIDE; i += 256) {
      const int r = i / (2 * ESPM_HS_STRIDE), j = i - r * 2 * ESPM_HS_STRIDE;
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, x.hstat_local[j >> 1]);
      const unsigned int half = (j & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
      __hip_atomic_store(gran(r, x.rank, 34 * x.nfl + j), ((unsigned long long)x.seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < x.world) xchg_store_flag(flag(threadIdx.x, x.rank), x.seq, x.release);
    return;
  }
  const int e = (int)blockIdx.x - 1, mm = e / a.k, kk = e - mm * a.k;
  const float* gt = a.g_t + (size_t)mm * a.n_pad;
  const float* av = a.a + (size_t)kk * a.n_pad;
  const float wo = a.w_old[e];
  const float fx = a.fixed_w ? a.fixed_w[e] : -1.f;
  float s = 0.f;
  for (int c0 = threadIdx.x; c0 < a.n; c0 += 8 * 256) {
    float gv[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * 256;
      gv[u] = c < a.n ? gt[c] : 0.f;
      vv[u] = c < a.n ? av[c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fmaf(gv[u], vv[u], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  const float gta_mine = ((s_wave[0] + s_wave[1]) + s_wave[2]) + s_wave[3];
  if (lane < x.world)   // this rank's value of entry e, to every rank
    __hip_atomic_store(gran(lane, x.rank, e), ((unsigned long long)x.seq << 32) | __float_as_uint(gta_mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // lanes 0..15: rank r's value of entry e; 16..31 / 32..47: the low / high half of rank r's row sum kk of its new H block
  const int r_of = lane & 15, what = lane >> 4;
  const bool polls = what < 3 && r_of < x.world;
  const int g_idx = what == 0 ? e : 34 * x.nfl + 2 * (ESPM_HS_ROWSUM + kk) + (what - 1);
  unsigned int got = 0;
  {
    const long long t0 = wall_clock64();
    for (;;) {
      const unsigned long long v = polls ? __hip_atomic_load(gran(x.rank, r_of, g_idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ((unsigned long long)x.seq << 32);
      const bool ok = (unsigned int)(v >> 32) == x.seq;
      got = (unsigned int)v;
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
      if (wall_clock64() - t0 > x.max_ticks) {
        if (!ok) atomicAdd(err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  float gta = 0.f;
  double rs = 0.0;
  for (int r = 0; r < x.world; ++r) {   // rank order: the same sums on every rank
    gta += __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)got, r));
    const unsigned long long lo = (unsigned int)__builtin_amdgcn_readlane((int)got, 16 + r), hi = (unsigned int)__builtin_amdgcn_readlane((int)got, 32 + r);
    rs += __builtin_bit_cast(double, (hi << 32) | lo);
  }
  if (lane == 0) {   // updates.py:58-60, :70-76
    const float num = wo * gta, den = a.colsum_g[mm] * (float)rs;
    float wn = fmaxf(num / den, a.log_shift);
    if (fx >= 0.f) wn = fx;
    a.w_new[e] = wn;
  }
  if (e == 0) {   // the global statistics of the new H (the NEXT launch reads them) and the neighbours' boundary rows' flag
    for (int hl = lane; hl < ((2 * ESPM_HS_STRIDE + 63) & ~63); hl += 64) {   // (64 halves of statistics per pass: two passes in the widest build)
    const bool p2 = hl < 2 * ESPM_HS_STRIDE;
    unsigned int hv[16];
    const long long t0 = wall_clock64();
    for (;;) {
      unsigned long long v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        v[r] = (p2 && r < x.world) ? __hip_atomic_load(gran(x.rank, r, 34 * x.nfl + hl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                   : ((unsigned long long)x.seq << 32);
      bool all = true;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        all = all && (unsigned int)(v[r] >> 32) == x.seq;
        hv[r] = (unsigned int)v[r];
      }
      if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
      if (wall_clock64() - t0 > x.max_ticks) {
        if (!all) atomicAdd(err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < x.world) {
        const unsigned int other = (unsigned int)__shfl_xor((int)hv[r], 1, 64);
        const unsigned long long bits = (lane & 1) ? (((unsigned long long)hv[r] << 32) | other) : (((unsigned long long)other << 32) | hv[r]);
        const double v2 = __builtin_bit_cast(double, bits);
        g = (hl >> 1) < ESPM_HS_MAX ? g + v2 : fmax(g, v2);
      }
    if (p2 && !(lane & 1)) x.hstat_out[hl >> 1] = g;
    }
    if (lane < x.world) xchg_wait_flag(flag(x.rank, lane), x.seq, x.max_ticks, err);
  }
}

// ---- the dictionary-G W step by COLUMNS: one launch behind the slab reduction (round 5) ------------------------------------------------
// With a dictionary of few columns (C5: G 1980 x 17, k = 8) component kk of the whole W update is small enough for ONE workgroup:
//   G^T A[:, kk]   m dot products over the n channels (a thread holds its 1 or 2 channels' rows of G^T in registers, and A[kk, :] once for all m),
//   (sharded) the m values cross the links as granules and come back summed in rank order, with the ranks' row sums of H'[kk, :],
//   W'[:, kk] = max(W (G^T A) / (colsum(G) rowsum(H')), eps), fixed_W                                        (updates.py:58-60, :70-76)
//   G W'[:, kk]    from the SAME registers of G^T: the column of the table the next H update gathers from, its sum as colsum(G)^T W'[:, kk].
// Nothing a workgroup needs comes from another workgroup of the launch, so what were two launches with a boundary between them - the update
// per entry of W (m k workgroups, each loading a row of A and a row of G^T for ONE dot product) and the rows of G W' (which need all of W') -
// is one: configuration 5's W step is two launches behind the fused one instead of three.  What IS global - mean(W') for
// rel_W (base.py:323) - is formed by one extra wave (the finisher) from the columns' sums, which arrive as 8-byte granules {launch nonce,
// fp32 sum}: value and "it is there" in one store, no counter to reset, stale content of the scratch never matches (the nonce is the
// process's launch count).  The entries of W' the finisher compares are stored write-through (sc1) and drained before the granule, and read
// with sc1 loads behind it (MI355X_MICROARCH.md, valid forms: data-tagged granules; sc1 payload drained before the flag, sc1 loads).
// Built for m <= GC_MMAX columns and n_pad <= GC_CPT * GC_THREADS channels; anything else keeps the launches above.
constexpr int GC_THREADS = 1024, GC_CPT = 2, GC_MMAX = 32;

// the extra workgroup of a sharded exchange launch: boundary rows of H' (16-byte write-through stores), this rank's statistics as granules, the rows' flag
__device__ __forceinline__ void gxchg_extra_role(const WGxchgArgs& x, int nthreads) {
  auto record = [&](int dst, int src_rank) { return x.mbox[dst] + x.slot_base + (size_t)src_rank * x.rec_bytes; };
  auto flag = [&](int dst, int src_rank) {
    return reinterpret_cast<unsigned int*>(x.mbox[dst] + x.wgflags_off + ((size_t)src_rank * x.nfl + (x.nfl - 1)) * sizeof(unsigned int));
  };
  const size_t GRAN = (size_t)34 * x.nfl + 2 * ESPM_HS_STRIDE;
  auto gran = [&](int dst, int src_rank, int g) {
    return reinterpret_cast<unsigned long long*>(x.mbox[dst] + x.gran_off) + (size_t)src_rank * GRAN + g;
  };
  if (x.with_halo) {
    for (int d = -1; d <= 1; ++d) {
      const int r = x.rank + d;
      if (r < 0 || r >= x.world) continue;
      float* top = reinterpret_cast<float*>(record(r, x.rank) + x.top_off);
      float* bot = reinterpret_cast<float*>(record(r, x.rank) + x.bot_off);
      if ((x.halo_ny & 3) == 0) {
        typedef float xf4 __attribute__((ext_vector_type(4)));
        for (int e = 4 * threadIdx.x; e < x.halo_k * x.halo_ny; e += 4 * nthreads) {
          const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
          const xf4 vt = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + j);
          const xf4 vb = *reinterpret_cast<const xf4*>(x.halo_h + (size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j);
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(top + e), "v"(vt) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(bot + e), "v"(vb) : "memory");
        }
      } else {
        for (int e = threadIdx.x; e < x.halo_k * x.halo_ny; e += nthreads) {
          const int kk = e / x.halo_ny, j = e - kk * x.halo_ny;
          __hip_atomic_store(top + e, x.halo_h[(size_t)kk * x.halo_ppad + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(bot + e, x.halo_h[(size_t)kk * x.halo_ppad + (size_t)(x.halo_nx - 1) * x.halo_ny + j], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  for (int i = threadIdx.x; i < x.world * 2 * ESPM_HS_STRIDE; i += nthreads) {
    const int r = i / (2 * ESPM_HS_STRIDE), j = i - r * 2 * ESPM_HS_STRIDE;
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, x.hstat_local[j >> 1]);
    const unsigned int half = (j & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
    __hip_atomic_store(gran(r, x.rank, 34 * x.nfl + j), ((unsigned long long)x.seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < x.world) xchg_store_flag(flag(threadIdx.x, x.rank), x.seq, x.release);
}

// one wave: the global statistics of the new H from the ranks' granules (the NEXT launch reads them), then the neighbours' boundary rows' flag
__device__ __forceinline__ void gxchg_global_stats_role(const WGxchgArgs& x, int lane) {
  auto flag = [&](int dst, int src_rank) {
    return reinterpret_cast<unsigned int*>(x.mbox[dst] + x.wgflags_off + ((size_t)src_rank * x.nfl + (x.nfl - 1)) * sizeof(unsigned int));
  };
  const size_t GRAN = (size_t)34 * x.nfl + 2 * ESPM_HS_STRIDE;
  auto gran = [&](int dst, int src_rank, int g) {
    return reinterpret_cast<unsigned long long*>(x.mbox[dst] + x.gran_off) + (size_t)src_rank * GRAN + g;
  };
  unsigned int* err = reinterpret_cast<unsigned int*>(x.mbox[x.rank] + x.err_off);
  for (int hl = lane; hl < ((2 * ESPM_HS_STRIDE + 63) & ~63); hl += 64) {   // (64 halves of statistics per pass: two passes in the widest build)
  const bool p2 = hl < 2 * ESPM_HS_STRIDE;
  unsigned int hv[16];
  const long long t0 = wall_clock64();
  for (;;) {
    unsigned long long v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      v[r] = (p2 && r < x.world) ? __hip_atomic_load(gran(x.rank, r, 34 * x.nfl + hl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                 : ((unsigned long long)x.seq << 32);
    bool all = true;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      all = all && (unsigned int)(v[r] >> 32) == x.seq;
      hv[r] = (unsigned int)v[r];
    }
    if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
    if (wall_clock64() - t0 > x.max_ticks) {
      if (!all) atomicAdd(err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  double g = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (r < x.world) {
      const unsigned int other = (unsigned int)__shfl_xor((int)hv[r], 1, 64);
      const unsigned long long bits = (lane & 1) ? (((unsigned long long)hv[r] << 32) | other) : (((unsigned long long)other << 32) | hv[r]);
      const double v2 = __builtin_bit_cast(double, bits);
      g = (hl >> 1) < ESPM_HS_MAX ? g + v2 : fmax(g, v2);
    }
  if (p2 && !(lane & 1)) x.hstat_out[hl >> 1] = g;
  }
  if (lane < x.world) xchg_wait_flag(flag(x.rank, lane), x.seq, x.max_ticks, err);
}

// grid: [XCHG: workgroup 0 = the exchange's extra workgroup] k column workgroups, then the finisher (one wave does the work)
template <bool XCHG>
__global__ __launch_bounds__(GC_THREADS) void w_gcol_kernel(const WGxchgArgs x, unsigned int nonce) {
  const WFinishArgs& a = x.f;
  __shared__ float s_red[GC_THREADS / 64][GC_MMAX];
  __shared__ unsigned int s_got[GC_MMAX + 2][16];
  __shared__ float s_w[GC_MMAX], s_csg[GC_MMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m, k = a.k, mk = m * k;
  const int first = XCHG ? 1 : 0;
  unsigned long long* colgran = reinterpret_cast<unsigned long long*>(a.scratch);   // [k] {nonce, fp32 sum of W'[:, kk]}
  if (XCHG && blockIdx.x == 0) {
    gxchg_extra_role(x, GC_THREADS);
    return;
  }
  if ((int)blockIdx.x == first + k) {   // ---- the finisher: mean(W'), rel_W (base.py:323); sharded: the global statistics of H'
    if (tid >= 64) return;
    if (XCHG) gxchg_global_stats_role(x, lane);
    if (lane < KP - k) a.colsum_gw[k + lane] = 0.0;
    if (!a.hist_slot) return;
    float sw = 0.f;
    const long long t0 = wall_clock64();
    for (;;) {
      const unsigned long long v = lane < k ? __hip_atomic_load(colgran + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)nonce << 32);
      const bool ok = (unsigned int)(v >> 32) == nonce;
      sw = __uint_as_float((unsigned int)v);
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
      if (wall_clock64() - t0 > 4 * x.max_ticks) break;   // (every column workgroup reaches its granule: its own waits are bounded)
      __builtin_amdgcn_s_sleep(1);
    }
    double sum = 0.0;
    for (int kk = 0; kk < k; ++kk) sum += (double)__uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)__float_as_uint(sw), kk));
    const double shift = (double)a.rel_tol * (sum / (double)mk);
    double rel = 0.0;
    for (int e = lane; e < mk; e += 64) {
      const double wn = (double)__hip_atomic_load(a.w_new + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), wov = (double)a.w_old[e];
      rel = fmax(rel, fabs(wn - wov) / (wn + shift));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    if (lane == 0) a.hist_slot[ESPM_HI_REL_W] = rel;
    return;
  }
  // ---- column kk ----
  const int kk = (int)blockIdx.x - first;
  float av[GC_CPT], gv[GC_MMAX][GC_CPT];
#pragma unroll
  for (int u = 0; u < GC_CPT; ++u) {
    const int c = tid + GC_THREADS * u;
    av[u] = c < a.n ? a.a[(size_t)kk * a.n_pad + c] : 0.f;
#pragma unroll
    for (int mm = 0; mm < GC_MMAX; ++mm) gv[mm][u] = (c < a.n && mm < m) ? a.g_t[(size_t)mm * a.n_pad + c] : 0.f;
  }
  float wo = 1.f, fx = -1.f;
  if (tid < m) {
    wo = a.w_old[tid * k + kk];
    if (a.fixed_w) fx = a.fixed_w[tid * k + kk];
    s_csg[tid] = a.colsum_g[tid];
  }
#pragma unroll
  for (int mm = 0; mm < GC_MMAX; ++mm) {
    if (mm < m) {   // (uniform)
      float part = gv[mm][0] * av[0];
#pragma unroll
      for (int u = 1; u < GC_CPT; ++u) part = fmaf(gv[mm][u], av[u], part);
      part = wave_sum(part);
      if (lane == 0) s_red[wave][mm] = part;
    }
  }
  __syncthreads();
  float gta = 0.f;
  if (tid < m) {
#pragma unroll
    for (int w = 0; w < GC_THREADS / 64; ++w) gta += s_red[w][tid];   // wave order
  }
  double rs;
  if constexpr (XCHG) {
    const size_t GRAN = (size_t)34 * x.nfl + 2 * ESPM_HS_STRIDE;
    auto gran = [&](int dst, int src_rank, int g) {
      return reinterpret_cast<unsigned long long*>(x.mbox[dst] + x.gran_off) + (size_t)src_rank * GRAN + g;
    };
    unsigned int* err = reinterpret_cast<unsigned int*>(x.mbox[x.rank] + x.err_off);
    if (tid < m) {   // this rank's value of entry (tid, kk), to every rank
      const unsigned long long g = ((unsigned long long)x.seq << 32) | __float_as_uint(gta);
      for (int r = 0; r < x.world; ++r) __hip_atomic_store(gran(r, x.rank, tid * k + kk), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // thread (row, r): rank r's value of entry row (< m), or (rows m, m + 1) the low / high half of rank r's row sum kk of its new H block
    const int row = tid >> 4, r_of = tid & 15;
    if (row < m + 2) {   // (whole waves: 16 threads per row, 4 rows per wave)
      const bool polls = r_of < x.world;
      const int g_idx = row < m ? row * k + kk : 34 * x.nfl + 2 * (ESPM_HS_ROWSUM + kk) + (row - m);
      unsigned int got = 0;
      const long long t0 = wall_clock64();
      for (;;) {
        const unsigned long long v = polls ? __hip_atomic_load(gran(x.rank, r_of, g_idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ((unsigned long long)x.seq << 32);
        const bool ok = (unsigned int)(v >> 32) == x.seq;
        got = (unsigned int)v;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        if (wall_clock64() - t0 > x.max_ticks) {
          if (!ok) atomicAdd(err, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      s_got[row][r_of] = got;
    }
    __syncthreads();
    gta = 0.f;
    rs = 0.0;
    for (int r = 0; r < x.world; ++r) {   // rank order: the same sums on every rank
      if (tid < m) gta += __uint_as_float(s_got[tid][r]);
      rs += __builtin_bit_cast(double, ((unsigned long long)s_got[m + 1][r] << 32) | s_got[m][r]);
    }
  } else {
    rs = a.hstat[ESPM_HS_ROWSUM + kk];
  }
  if (tid < m) {   // updates.py:58-60, :70-76
    const float num = wo * gta, den = s_csg[tid] * (float)rs;
    float wn = fmaxf(num / den, a.log_shift);
    if (fx >= 0.f) wn = fx;
    __hip_atomic_store(a.w_new + tid * k + kk, wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (sc1: the finisher reads it in this launch)
    s_w[tid] = wn;
  }
  __syncthreads();
  // the column of G W' (updates.py:107 of the next half step), stored / xscale with a positive floor; padding channels: X = 0 there
  const float inv_scale = 1.f / a.xscale;
#pragma unroll
  for (int u = 0; u < GC_CPT; ++u) {
    const int c = tid + GC_THREADS * u;
    if (c < a.n_pad) {
      float v = 1.f;
      if (c < a.n) {
        float row = 0.f;
#pragma unroll
        for (int mm = 0; mm < GC_MMAX; ++mm)
          if (mm < m) row = fmaf(gv[mm][u], s_w[mm], row);
        v = fmaxf(row, a.gw_floor) * inv_scale;
      }
      a.gw_s[(size_t)c * KP + kk] = v;
    }
  }
  if (wave == 0) {   // (the threads that stored W'[:, kk] are this wave's: its drain covers them)
    double cs = 0.0;
    float swf = 0.f;
    if (lane == 0) {
      for (int mm = 0; mm < m; ++mm) {
        cs += (double)s_csg[mm] * (double)s_w[mm];   // colsum(G W')[kk] = colsum(G)^T W'[:, kk] (round 4: DESIGN.md section 2)
        swf += s_w[mm];
      }
      a.colsum_gw[kk] = cs;
    }
    if (a.hist_slot) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(colgran + kk, ((unsigned long long)nonce << 32) | __float_as_uint(swf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static bool w_gcol_enabled() {
  static const bool on = [] {
    const char* e = getenv("ESPM_W_GCOL");
    return !(e && e[0] == '0');
  }();
  return on;
}
bool w_gsplit_applies(const WFinishArgs& args);
// the column form applies where the two-launch form does, the dictionary has at most GC_MMAX columns and a thread holds its channels
bool w_gcol_applies(const WFinishArgs& args) {
  return w_gsplit_applies(args) && args.m <= GC_MMAX && args.n_pad <= GC_CPT * GC_THREADS && args.k <= KP && (size_t)2 * args.m * args.k * sizeof(float) >= (size_t)args.k * 8 &&
         w_gcol_enabled();
}
static unsigned int w_gcol_nonce() {
  static std::atomic<unsigned int> count{0};
  unsigned int v = ++count;
  if (v == 0) v = ++count;   // (0 is what a zeroed scratch holds)
  return v;
}
int launch_w_gcol(const WFinishArgs& f, hipStream_t stream) {
  WGxchgArgs x = {};
  x.f = f;
  x.world = 1;
  x.max_ticks = 200000000LL;
  hipLaunchKernelGGL(w_gcol_kernel<false>, dim3(f.k + 1), dim3(GC_THREADS), 0, stream, x, w_gcol_nonce());
  return check_hip(hipGetLastError(), "w_finish (dictionary G, by columns)");
}

static bool w_gsplit_enabled();
bool w_gsplit_applies(const WFinishArgs& args) {
  const long mk = (long)args.m * args.k;
  const int nwg_b = (args.n_pad + 255) / 256;
  return args.g && args.g_t && args.m > 0 && args.update_w && !args.simplex_w && args.pg_gamma_w <= 0.f && !args.breg_sr && mk <= 8192 && args.scratch &&
         (size_t)2 * mk * sizeof(float) >= (size_t)nwg_b * KP * sizeof(double) + 16 && w_gsplit_enabled();
}
int launch_w_gfinish_gw(const WFinishArgs& args, hipStream_t stream);

int launch_w_gxchg_update(const WFinishArgs& f, const espm_xchg* xc, unsigned int seq, const double* hstat_local, double* hstat_out,
                          const float* h_new, int nx, int ny, int p_pad, int with_halo, hipStream_t stream) {
  WGxchgArgs x;
  x.f = f;
  const int mk = f.m * f.k;
  ESPM_REQUIRE((size_t)mk <= (size_t)34 * xc->wgflags, "exchange: %d entries of G^T A, the mailbox holds granules for %d", mk, 34 * xc->wgflags);
  for (int r = 0; r < 16; ++r) x.mbox[r] = r < xc->world ? xc->peers[r] : nullptr;
  for (int r = 0; r < xc->world; ++r) ESPM_REQUIRE(x.mbox[r], "exchange: rank %d is not connected (espm_xchg_connect)", r);
  x.hstat_local = hstat_local;
  x.hstat_out = hstat_out;
  x.world = xc->world;
  x.rank = xc->rank;
  x.nfl = xc->wgflags;
  x.with_halo = with_halo;
  x.rec_bytes = xc->record_bytes;
  x.slot_base = (size_t)(seq & 1u) * xc->world * xc->record_bytes;
  x.wgflags_off = xc->off_wgflags;
  x.gran_off = xc->off_gran + (size_t)(seq & 1u) * xc->world * ((size_t)34 * xc->wgflags + 2 * ESPM_HS_STRIDE) * sizeof(unsigned long long);
  x.err_off = xc->off_err;
  const size_t hstat_off = (size_t)f.k * f.n_pad * 4;
  x.top_off = hstat_off + ESPM_HS_STRIDE * 8;
  x.bot_off = x.top_off + (size_t)f.k * (ny > 0 ? ny : 0) * 4;
  x.seq = seq;
  x.max_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  x.release = xc->order;
  x.halo_h = h_new;
  x.halo_k = f.k;
  x.halo_nx = nx;
  x.halo_ny = ny;
  x.halo_ppad = p_pad;
  if (w_gcol_applies(f)) {   // one launch: G^T A by columns, the exchange, W', the columns of G W' (w_gcol_kernel)
    hipLaunchKernelGGL(w_gcol_kernel<true>, dim3(f.k + 2), dim3(GC_THREADS), 0, stream, x, w_gcol_nonce());
    return check_hip(hipGetLastError(), "w_gcol (exchange) launch");
  }
  hipLaunchKernelGGL(w_gxchg_update_kernel, dim3(mk + 1), dim3(256), 0, stream, x);
  if (int rc = check_hip(hipGetLastError(), "w_gxchg_update launch")) return rc;
  return launch_w_gfinish_gw(f, stream);
}

static bool w_gsplit_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("ESPM_W_GSPLIT");
    on = !(e && e[0] == '0');
  }
  return on != 0;
}

int launch_w_gfinish_gw(const WFinishArgs& args, hipStream_t stream) {
  const int nwg_b = (args.n_pad + 255) / 256;
  hipLaunchKernelGGL(w_gfinish_gw_kernel, dim3(nwg_b + 1), dim3(256), (size_t)args.m * args.k * sizeof(float), stream, args);
  return check_hip(hipGetLastError(), "w_finish (dictionary G: rows of G W')");
}

int launch_w_finish(const WFinishArgs& args, hipStream_t stream) {
  const int M = args.m > 0 ? args.m : args.n;
  const long mk = (long)M * args.k;
  // dictionary G, W' = max(W (G^T A) / (colsum G rowsum H'), eps): the two-launch finish above
  // (scratch: w_scratch's 2 m k floats must hold the second launch's partial column sums and its ticket - zero between launches)
  if (w_gcol_applies(args)) return launch_w_gcol(args, stream);
  if (w_gsplit_applies(args)) {
    hipLaunchKernelGGL(w_gfinish_update_kernel, dim3((unsigned)mk), dim3(256), 0, stream, args);
    if (int rc = check_hip(hipGetLastError(), "w_finish (dictionary G: update)")) return rc;
    return launch_w_gfinish_gw(args, stream);
  }
  const int span = M > args.n_cm ? M : args.n_cm;
  // G = identity, up to 8 components (the narrow build), up to 2048 rows: 8 waves with 256 registers each (with the simplex
  // over W at the headline size 49 -> 37 us at k = 5, iteration 278 -> 248 us at k = 8); a dictionary G keeps the 16 waves
  // (its loops over the rows of G want them: C5 141 vs 151 us)
  const int nt = (!args.g && args.k <= WF_HALF_MAX_K && span <= 4 * 512) ? WF_FEW_THREADS : WF_THREADS;
  const int crows = (span + nt - 1) / nt;               // channels (or, with G = identity, rows of W) per thread
  const int rows = args.g ? (M + nt - 1) / nt : crows;   // rows of W per thread
  // G given: [M][k rounded up to 4] new W, [M k] G^T A, and the per-wave partials of the all-threads G^T A (within the 64 KB a
  // kernel gets without asking)
  const size_t lds = args.g ? ((size_t)M * ((args.k + 3) / 4 * 4) + (size_t)mk * (1 + (args.g_t && mk <= WF_GTA_PAR ? WF_THREADS / 64 : 0))) * sizeof(float) : 0;
  // (the widest build - 17..32 components - has the general one-workgroup finish only: the register-resident ones are not built there;
  //  ESPM_W_FINISH_GENERAL=1 sends the other builds there too - tests: images small enough for the register-resident kernels never reach it)
  static const bool general_only = [] { const char* e = getenv("ESPM_W_FINISH_GENERAL"); return e && e[0] == '1'; }();
  if (KP <= 16 && !general_only && crows <= (nt == 256 ? 8 : 4) && rows <= (nt == 256 ? 8 : 4) && (!args.g || (mk <= WF_GTA_MAX && lds <= 64 * 1024))) {
    switch (args.k) {
#if ESPM_KP <= 16
#define ESPM_X(KK)                                                              \
  case KK:                                                                      \
    if (nt == WF_FEW_THREADS) launch_fast<KK, (KK <= WF_HALF_MAX_K ? WF_FEW_THREADS : WF_THREADS)>(args, rows, crows, lds, stream); \
    else launch_fast<KK, WF_THREADS>(args, rows, crows, lds, stream);                  \
    break;
      ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
      default: return set_error(ESPM_EUNSUPPORTED, "w_finish: k=%d not built", args.k);
    }
  } else {
    hipLaunchKernelGGL(w_finish_kernel, dim3(1), dim3(WF_THREADS), 0, stream, args);
  }
  return check_hip(hipGetLastError(), "w_finish launch");
}

}  // namespace espm

#ifdef ESPM_PHASE_CLOCK
// debug build only (tools/analysis/w_finish_clock.py): where this file's kernels write their phase stamps
extern "C" int espm_debug_phase_buffer_w(void* dev_ptr) {
  unsigned long long* p = static_cast<unsigned long long*>(dev_ptr);
  return espm::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(espm::espm_phase_buf), &p, sizeof(p)), "phase buffer (w)");
}
#endif
