// The W accumulation kernel template (see mu_w_step.hip); in a header for the tuning harness.
#pragma once
#include "mu_common.hpp"

namespace espm {

// K components, XT storage type, CH channels per lane (one 16-byte load), UP pixels per load group,
// PIPE: two groups in flight in two register sets.  A workgroup = 4 waves = 256 * CH channels,
// blockIdx.y walks further channel chunks, blockIdx.x the pixel blocks.
template <int K, typename XT, int CH, int UP, bool PIPE>
__global__ __launch_bounds__(256) void w_accum_kernel(const WAccumArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c_raw = ((blockIdx.y * 4 + wave) * 64 + lane) * CH;
  const bool active = c_raw < a.n_pad;
  // lanes past the last channel read (and discard) the last valid vector: no predicated loads
  const int c0 = active ? c_raw : a.n_pad - CH;

  float gw[CH][K];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) gw[i][kk] = a.gw_s[(size_t)(c0 + i) * KP + kk];

  float acc[CH][K];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[i][kk] = 0.f;

  const int j_begin = blockIdx.x * a.ppb;
  const int j_end = min(a.p, j_begin + a.ppb);
  const XT* xbase = static_cast<const XT*>(a.x_pm) + c0;

  // one pixel: Y = GW H[:, j], R = X / Y, A += R H[:, j]^T   (updates.py:38-39, :53, :59)
  auto pixel = [&](const XVec<XT, CH>& xv, const float (&hk)[K]) {
    float x[CH];
    xv.get(x);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float y = gw[i][0] * hk[0];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = fmaf(gw[i][kk], hk[kk], y);
      const float r = x[i] * __builtin_amdgcn_rcpf(y);
#pragma unroll
      for (int kk = 0; kk < K; ++kk) acc[i][kk] = fmaf(r, hk[kk], acc[i][kk]);
    }
  };

  struct Group {
    XVec<XT, CH> x[UP];
    float h[UP][K];
  };
  auto load_group = [&](Group& grp, int j) {
    const XT* xr = xbase + (size_t)j * a.n_pad;
    const float* hr = a.h_t + (size_t)j * KP;  // wave-uniform -> scalar loads
#pragma unroll
    for (int u = 0; u < UP; ++u) grp.x[u].load(xr + (size_t)u * a.n_pad);
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) grp.h[u][kk] = hr[u * KP + kk];
  };
  auto compute_group = [&](const Group& grp) {
#pragma unroll
    for (int u = 0; u < UP; ++u) pixel(grp.x[u], grp.h[u]);
  };

  int j = j_begin;
  if constexpr (PIPE) {
    const int ngroups = (j_end - j_begin) / UP;
    if (ngroups > 0) {
      const int j_last = j_begin + (ngroups - 1) * UP;
      Group ga, gb;
      load_group(ga, j);
      int g = 0;
      for (; g + 2 <= ngroups; g += 2) {
        load_group(gb, min(j + UP, j_last));
        compute_group(ga);
        load_group(ga, min(j + 2 * UP, j_last));
        compute_group(gb);
        j += 2 * UP;
      }
      if (g < ngroups) {
        compute_group(ga);
        j += UP;
      }
    }
  } else {
    for (; j + UP <= j_end; j += UP) {
      Group grp;
      load_group(grp, j);
      compute_group(grp);
    }
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
      for (int i = 0; i < CH; ++i) dst[i] = acc[i][kk];
    }
  }
}

}  // namespace espm
