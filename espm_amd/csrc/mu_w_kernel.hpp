// The W accumulation kernel template (see mu_w_step.hip); in a header for the tuning harness.
#pragma once
#include "mu_common.hpp"

namespace espm {

// K components, XT storage type, CH channels per lane (one 16-byte load), UP pixels per load group,
// NBUF: depth of the register ring of X load groups kept in flight.  A workgroup = 4 waves = 256 * CH channels,
// blockIdx.y walks further channel chunks, blockIdx.x the pixel blocks.
// L2: the Frobenius branch (updates.py:31-36): A = X H^T, no ratio.
template <int K, typename XT, int CH, int UP, int NBUF, bool L2 = false>
__global__ __launch_bounds__(256) void w_accum_kernel(const WAccumArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c_raw = ((blockIdx.y * 4 + wave) * 64 + lane) * CH;
  const bool active = c_raw < a.n_pad;
  // lanes past the last channel read (and discard) the last valid vector: no predicated loads
  const int c0 = active ? c_raw : a.n_pad - CH;

  // channel PAIRS in fp32x2 registers (v_pk_fma_f32 with the wave-uniform H entry broadcast from an
  // SGPR: two FMAs per issue slot, see mu_h_kernel.hpp)
  constexpr int C2 = CH / 2;
  f2 gw[C2][K];
#pragma unroll
  for (int i = 0; i < C2; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
      gw[i][kk] = f2{a.gw_s[(size_t)(c0 + 2 * i) * KP + kk], a.gw_s[(size_t)(c0 + 2 * i + 1) * KP + kk]};

  f2 acc[C2][K];
#pragma unroll
  for (int i = 0; i < C2; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[i][kk] = f2{0.f, 0.f};

  const int j_begin = blockIdx.x * a.ppb;
  const int j_end = min(a.p, j_begin + a.ppb);
  const XT* xbase = static_cast<const XT*>(a.x_pm) + c0;

  // one pixel: Y = GW H[:, j], R = X / Y, A += R H[:, j]^T   (updates.py:38-39, :53, :59)
  auto pixel = [&](const XVec<XT, CH>& xv, const float (&hk)[K]) {
    f2 x[C2];
    xv.get2(x);
#pragma unroll
    for (int i = 0; i < C2; ++i) {
      f2 y = hk[0] * gw[i][0];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = hk[kk] * gw[i][kk] + y;
      const f2 r = L2 ? x[i] : x[i] * f2{__builtin_amdgcn_rcpf(y.x), __builtin_amdgcn_rcpf(y.y)};
#pragma unroll
      for (int kk = 0; kk < K; ++kk) acc[i][kk] = hk[kk] * r + acc[i][kk];
    }
  };

  // X rows are requested NBUF groups of UP pixels ahead of their use (register ring), the wave-uniform
  // H columns (scalar cache) one group ahead; prefetch addresses are clamped to the last full group.
  struct XGroup {
    XVec<XT, CH> x[UP];
  };
  struct HGroup {
    float h[UP][K];
  };
  auto load_x = [&](XGroup& grp, int j) {
    const XT* xr = xbase + (size_t)j * a.n_pad;
#pragma unroll
    for (int u = 0; u < UP; ++u) grp.x[u].load(xr + (size_t)u * a.n_pad);
  };
  auto load_h = [&](HGroup& grp, int j) {
    const float* hr = a.h_t + (size_t)j * KP;  // wave-uniform -> scalar loads
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) grp.h[u][kk] = hr[u * KP + kk];
  };
  auto compute_group = [&](const XGroup& xg, const HGroup& hg) {
#pragma unroll
    for (int u = 0; u < UP; ++u) pixel(xg.x[u], hg.h[u]);
  };

  int j = j_begin;
  if constexpr (NBUF >= 2) {
    const int ngroups = (j_end - j_begin) / UP;
    if (ngroups > 0) {
      const int j_last = j_begin + (ngroups - 1) * UP;
      XGroup xs[NBUF];
      HGroup ha, hb;
#pragma unroll
      for (int b = 0; b < NBUF; ++b) load_x(xs[b], min(j + b * UP, j_last));
      load_h(ha, j);
      int g = 0;
      for (; g + 2 * NBUF <= ngroups; g += 2 * NBUF) {
#pragma unroll
        for (int b = 0; b < 2 * NBUF; ++b) {
          HGroup& cur = (b & 1) ? hb : ha;
          HGroup& nxt = (b & 1) ? ha : hb;
          load_h(nxt, min(j + UP, j_last));
          compute_group(xs[b % NBUF], cur);
          load_x(xs[b % NBUF], min(j + NBUF * UP, j_last));
          j += UP;
        }
      }
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        if (g < ngroups) {
          HGroup hl;
          load_h(hl, j);
          compute_group(xs[b], hl);
          j += UP;
          ++g;
        }
      }
    }
  }
  for (; j + UP <= j_end; j += UP) {
    XGroup xg;
    HGroup hg;
    load_x(xg, j);
    load_h(hg, j);
    compute_group(xg, hg);
  }
  for (; j < j_end; ++j) {
    XVec<XT, CH> xv;
    xv.load(xbase + (size_t)j * a.n_pad);
    float hk[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) hk[kk] = a.h_t[(size_t)j * KP + kk];
    pixel(xv, hk);
  }

  if (active) {
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      float* dst = a.a_slab + ((size_t)blockIdx.x * K + kk) * a.n_pad + c0;
#pragma unroll
      for (int i = 0; i < C2; ++i) {
        dst[2 * i] = acc[i][kk].x;
        dst[2 * i + 1] = acc[i][kk].y;
      }
    }
  }
}

}  // namespace espm
