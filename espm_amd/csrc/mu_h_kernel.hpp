// The H-step kernel template (see mu_h_step.hip for the description); in a header so that the tuning
// harness (tools/tune) can instantiate variants next to the product's dispatch table.
#pragma once
#include "mu_common.hpp"

namespace espm {

// K components, XT storage type of X, PX pixels per lane (tile = 64 * PX pixels), NW waves per
// workgroup (they split the channel range), LOSS: accumulate the KL term, U channels per load group,
// PIPE: explicit register double buffering of the X loads and GW rows (two groups in flight).
template <int K, typename XT, int PX, int NW, bool LOSS, int U, bool PIPE>
__global__ __launch_bounds__(NW * 64) void h_step_kernel(const HStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][K][TP]
  constexpr int TP = 64 * PX;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * TP;
  const int px0 = tile0 + lane * PX;

  float h[K][PX];
#pragma unroll
  for (int kk = 0; kk < K; ++kk) load_f32<PX>(a.h_in + (size_t)kk * a.p_pad + px0, h[kk]);

  float num[K][PX];
  float kl[PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    kl[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) num[kk][i] = 0.f;
  }

  const int chunk = (a.n + NW - 1) / NW;
  const int c_begin = min(a.n, wave * chunk);
  const int c_end = min(a.n, c_begin + chunk);
  const XT* xbase = static_cast<const XT*>(a.x_cm) + px0;

  // one channel: Y = GW[c,:] H, R = X / Y, num += GW[c,:]^T R   (updates.py:127-128)
  auto channel = [&](const XVec<XT, PX>& xv, const float (&gk)[K]) {
    float x[PX];
    xv.get(x);
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      float y = gk[0] * h[0][i];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = fmaf(gk[kk], h[kk][i], y);
      // R = X / Y; with the loss the tiny offset keeps log2(R) finite where X = 0 (0 * finite = 0) at
      // no extra cost (it rides in the fma) and is far below fp32 resolution of any non-zero R
      const float r = LOSS ? fmaf(x[i], __builtin_amdgcn_rcpf(y), 1e-37f) : x[i] * __builtin_amdgcn_rcpf(y);
#pragma unroll
      for (int kk = 0; kk < K; ++kk) num[kk][i] = fmaf(gk[kk], r, num[kk][i]);
      if constexpr (LOSS) kl[i] = fmaf(x[i], __builtin_amdgcn_logf(r), kl[i]);
    }
  };

  struct Group {
    XVec<XT, PX> x[U];
    float g[U][K];
  };
  auto load_group = [&](Group& grp, int c) {
    const XT* xr = xbase + (size_t)c * a.p_pad;
    const float* gr = a.gw_s + (size_t)c * KP;  // wave-uniform -> scalar loads
#pragma unroll
    for (int u = 0; u < U; ++u) grp.x[u].load(xr + (size_t)u * a.p_pad);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) grp.g[u][kk] = gr[u * KP + kk];
  };
  auto compute_group = [&](const Group& grp) {
#pragma unroll
    for (int u = 0; u < U; ++u) channel(grp.x[u], grp.g[u]);
  };

  int c = c_begin;
  if constexpr (PIPE) {
    // groups [c, c+U) and [c+U, c+2U) alternate between two register sets; the prefetch address is
    // clamped to the last full group of the chunk, so no load is predicated and none leaves the
    // wave's channel range (a clamped re-load is simply not consumed)
    const int ngroups = (c_end - c_begin) / U;
    if (ngroups > 0) {
      const int c_last = c_begin + (ngroups - 1) * U;
      Group ga, gb;
      load_group(ga, c);
      int g = 0;
      for (; g + 2 <= ngroups; g += 2) {
        load_group(gb, min(c + U, c_last));
        compute_group(ga);
        load_group(ga, min(c + 2 * U, c_last));
        compute_group(gb);
        c += 2 * U;
      }
      if (g < ngroups) {
        compute_group(ga);
        c += U;
      }
    }
  } else {
    for (; c + U <= c_end; c += U) {
      Group grp;
      load_group(grp, c);
      compute_group(grp);
    }
  }
  for (; c < c_end; ++c) {  // remainder channels one at a time
    XVec<XT, PX> xv;
    xv.load(xbase + (size_t)c * a.p_pad);
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
    for (int i = 0; i < PX; ++i) dst[i] = num[kk][i];
  }
  constexpr int NRED = ESPM_HP_NSCALAR + 2 * K;  // sums: scalars (but RELH) + K row sums; max: RELH + K row maxima
  double red[NRED];
#pragma unroll
  for (int i = 0; i < NRED; ++i) red[i] = 0.0;
  // layout inside red[]: [0..3] KL, REG, LAP, BAD (sums), [4..4+K) row sums, [4+K] RELH, [5+K..5+2K) maxima
  constexpr int R_ROWSUM = 4, R_RELH = 4 + K, R_MAX = 5 + K;
  if constexpr (LOSS) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PX; ++i) s += kl[i];
    red[ESPM_HP_KL] = (double)s;
  }
  __syncthreads();

  // ---- epilogue: one thread per pixel -------------------------------------------------------
  const float ls = a.lambda_l * a.sigma_l;
  float rel_shift = 0.f;
  if (a.have_prev) {  // base.py:324: tol * mean(H) of the state being evaluated (global row sums)
    double tot = 0.0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) tot += a.hstat_in[ESPM_HS_ROWSUM + kk];
    rel_shift = (float)((double)a.rel_tol * tot * a.inv_count);
  }
  for (int jj = threadIdx.x; jj < TP; jj += NW * 64) {
    const int q = tile0 + jj;
    if (q >= a.p) continue;
    float hin[K], nv[K], dv[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += smem[((size_t)w * K + kk) * TP + jj];
      hin[kk] = a.h_in[(size_t)kk * a.p_pad + q];
      nv[kk] = s * a.xscale;
      dv[kk] = (float)a.colsum_gw[kk];
    }
    if (a.have_prev) {
      // rel_H of the update that produced h_in: the other buffer still holds the previous H
      // (each thread reads its own entries before overwriting them below), base.py:324
      float worst = 0.f;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float hp = a.h_out[(size_t)kk * a.p_pad + q];
        worst = fmaxf(worst, fabsf(hin[kk] - hp) / (hin[kk] + rel_shift));
      }
      red[R_RELH] = fmax(red[R_RELH], (double)worst);
    }
    if (a.mu) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float m = a.mu[kk];
        dv[kk] += m / (hin[kk] + a.eps_reg);                          // updates.py:134-137
        red[ESPM_HP_REG] += (double)(m * logf(hin[kk] + a.eps_reg));  // measures.py:543-548
      }
    }
    if (a.lambda_l != 0.f) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float hl = a.grid_mode
                             ? stencil_hl(a.h_in + (size_t)kk * a.p_pad,
                                          a.halo_top ? a.halo_top + (size_t)kk * a.ny : nullptr,
                                          a.halo_bot ? a.halo_bot + (size_t)kk * a.ny : nullptr, q, a.nx, a.ny,
                                          hin[kk])
                             : hin[kk];
        const float mh = (float)a.hstat_in[ESPM_HS_MAX + kk];   // GLOBAL max over pixels, updates.py:139
        nv[kk] += ls * mh;                                      // updates.py:140
        dv[kk] += ls * mh + a.lambda_l * hl;                    // updates.py:141
        red[ESPM_HP_LAP] += (double)(hin[kk] * hl);             // measures.py:574-577
      }
    }
    if (!a.write_h) continue;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) nv[kk] *= hin[kk];          // updates.py:142
    if (a.simplex_h) {
      float delta, e[K];
      if (!simplex_root<float, K>(nv, dv, K, a.log_shift, fminf(a.tol, 1e-6f), 100, delta, e)) red[ESPM_HP_BAD] += 1.0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) dv[kk] = e[kk] + delta;  // = den + nu, formed without cancellation
    }
    float ht[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) ht[kk] = 0.f;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      float hn = fmaxf(nv[kk] / dv[kk], a.log_shift);          // updates.py:152
      if (a.fixed_h) {
        const float f = a.fixed_h[(size_t)kk * a.p_pad + q];
        if (f >= 0.f) hn = f;                                   // updates.py:154-155
      }
      if (!(hn <= 3.0e38f)) red[ESPM_HP_BAD] += 1.0;            // NaN or inf
      a.h_out[(size_t)kk * a.p_pad + q] = hn;
      ht[kk] = hn;
      red[R_ROWSUM + kk] += (double)hn;
      red[R_MAX + kk] = fmax(red[R_MAX + kk], (double)hn);
    }
    float4* dst = reinterpret_cast<float4*>(a.h_t + (size_t)q * KP);
    dst[0] = make_float4(ht[0], ht[1], ht[2], ht[3]);
    dst[1] = make_float4(ht[4], ht[5], ht[6], ht[7]);
  }

  __syncthreads();  // smem is reused as reduction scratch
  block_reduce<NRED, R_RELH>(red, reinterpret_cast<double*>(smem));
  if (threadIdx.x == 0) {
    // field-major records: hpart[field][block], so that the finalize kernel reads them coalesced
    double* out = a.hpart + blockIdx.x;
    const size_t nb = gridDim.x;
    out[ESPM_HP_KL * nb] = red[ESPM_HP_KL];
    out[ESPM_HP_REG * nb] = red[ESPM_HP_REG];
    out[ESPM_HP_LAP * nb] = red[ESPM_HP_LAP];
    out[ESPM_HP_BAD * nb] = red[ESPM_HP_BAD];
    out[ESPM_HP_RELH * nb] = red[R_RELH];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      out[(ESPM_HP_ROWSUM + kk) * nb] = kk < K ? red[R_ROWSUM + kk] : 0.0;
      out[(ESPM_HP_MAX + kk) * nb] = kk < K ? red[R_MAX + kk] : 0.0;
    }
  }
}

}  // namespace espm
