// H update of the SmoothNMF multiplicative rule, fused over one pixel tile per workgroup.
//
// Restates espm/estimators/updates.py:83-156 (KL branch) without materialising any (n, p)
// intermediate:  Y = GW H,  R = X / Y,  num = GW^T R  are formed per pixel column in registers,
// with lanes <-> pixels (so the contraction over channels needs no cross-lane traffic) and the
// channel range split over the waves of the workgroup.  X is streamed once, channel-major, with
// 16-byte (bf16 x 8) coalesced loads; GW rows are wave-uniform and come through the scalar cache.
// The epilogue (one thread per pixel) adds the log-sparsity and Laplacian terms, solves the
// per-pixel simplex multiplier (dicotomy.py:4-55) in registers, clamps and writes H', its
// transposed copy for the W-step, and per-workgroup partial sums:
//   - KL data term of the INPUT state: sum X log2(X / Y)   (measures.py:493-504, base.py:200-203)
//   - log regulariser and Laplacian quadratic form of the input state (measures.py:543-548, :574-577)
//   - row sums / row maxima of H' (updates.py:60, :139 of the NEXT half steps)
#include "mu_common.hpp"

namespace espm {

template <int K, typename XT, int PX, int NW, bool LOSS>
__global__ __launch_bounds__(NW * 64) void h_step_kernel(const HStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][K][TP]
  constexpr int TP = 64 * PX;
  constexpr int U = PX >= 8 ? 8 : 16;
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
  const int c_begin = wave * chunk;
  const int c_end = min(a.n, c_begin + chunk);
  const XT* xrow = static_cast<const XT*>(a.x_cm) + (size_t)c_begin * a.p_pad + px0;
  const float* g = a.gw_s + (size_t)c_begin * KP;

  auto body = [&](const XVec<XT, PX>& xv, const float* gc) {
    float gk[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) gk[kk] = gc[kk];  // wave-uniform -> scalar loads
    float x[PX];
    xv.get(x);
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      float y = gk[0] * h[0][i];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = fmaf(gk[kk], h[kk][i], y);
      const float r = x[i] * __builtin_amdgcn_rcpf(y);
#pragma unroll
      for (int kk = 0; kk < K; ++kk) num[kk][i] = fmaf(gk[kk], r, num[kk][i]);
      if constexpr (LOSS) kl[i] = fmaf(x[i], __builtin_amdgcn_logf(fmaxf(r, 1e-37f)), kl[i]);
    }
  };

  int c = c_begin;
  for (; c + U <= c_end; c += U) {
    XVec<XT, PX> xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u].load(xrow + (size_t)u * a.p_pad);
#pragma unroll
    for (int u = 0; u < U; ++u) body(xv[u], g + u * KP);
    xrow += (size_t)U * a.p_pad;
    g += U * KP;
  }
  for (; c < c_end; ++c) {
    XVec<XT, PX> xv;
    xv.load(xrow);
    body(xv, g);
    xrow += a.p_pad;
    g += KP;
  }

  // ---- cross-wave reduction of the numerators through LDS ---------------------------------
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    float* dst = smem + ((size_t)wave * K + kk) * TP + lane * PX;
#pragma unroll
    for (int i = 0; i < PX; ++i) dst[i] = num[kk][i];
  }
  double red[4 + 2 * K];
#pragma unroll
  for (int i = 0; i < 4 + 2 * K; ++i) red[i] = 0.0;
  if constexpr (LOSS) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PX; ++i) s += kl[i];
    red[ESPM_HP_KL] = (double)s;
  }
  __syncthreads();

  // ---- epilogue: one thread per pixel -------------------------------------------------------
  const float ls = a.lambda_l * a.sigma_l;
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
    if (a.mu) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float m = a.mu[kk];
        dv[kk] += m / (hin[kk] + a.eps_reg);                    // updates.py:134-137
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
      float hn = fmaxf(nv[kk] / dv[kk], a.log_shift);   // updates.py:152
      if (a.fixed_h) {
        const float f = a.fixed_h[(size_t)kk * a.p_pad + q];
        if (f >= 0.f) hn = f;                                   // updates.py:154-155
      }
      if (!(hn <= 3.0e38f)) red[ESPM_HP_BAD] += 1.0;            // NaN or inf
      a.h_out[(size_t)kk * a.p_pad + q] = hn;
      ht[kk] = hn;
      red[4 + kk] += (double)hn;
      red[4 + K + kk] = fmax(red[4 + K + kk], (double)hn);
    }
    float4* dst = reinterpret_cast<float4*>(a.h_t + (size_t)q * KP);
    dst[0] = make_float4(ht[0], ht[1], ht[2], ht[3]);
    dst[1] = make_float4(ht[4], ht[5], ht[6], ht[7]);
  }

  __syncthreads();  // smem is reused as reduction scratch
  block_reduce<4 + 2 * K, 4 + K>(red, reinterpret_cast<double*>(smem));
  if (threadIdx.x == 0) {
    double* out = a.hpart + (size_t)blockIdx.x * ESPM_HP_STRIDE;
    out[ESPM_HP_KL] = red[ESPM_HP_KL];
    out[ESPM_HP_REG] = red[ESPM_HP_REG];
    out[ESPM_HP_LAP] = red[ESPM_HP_LAP];
    out[ESPM_HP_BAD] = red[ESPM_HP_BAD];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      out[ESPM_HP_ROWSUM + kk] = kk < K ? red[4 + kk] : 0.0;
      out[ESPM_HP_MAX + kk] = kk < K ? red[4 + K + kk] : 0.0;
    }
  }
}

// ---- reduction of the per-workgroup records (one workgroup) ---------------------------------
__global__ __launch_bounds__(256) void h_finalize_kernel(const HFinalizeArgs a) {
  __shared__ double scratch[4 * (4 + 2 * KP)];
  double v[4 + 2 * KP];
#pragma unroll
  for (int i = 0; i < 4 + 2 * KP; ++i) v[i] = 0.0;
  for (int b = threadIdx.x; b < a.nblk; b += blockDim.x) {
    const double* r = a.hpart + (size_t)b * ESPM_HP_STRIDE;
#pragma unroll
    for (int i = 0; i < 4 + KP; ++i) v[i] += r[i];
#pragma unroll
    for (int i = 0; i < KP; ++i) v[4 + KP + i] = fmax(v[4 + KP + i], r[ESPM_HP_MAX + i]);
  }
  block_reduce<4 + 2 * KP, 4 + KP>(v, scratch);
  if (threadIdx.x == 0) {
    double sumy = 0.0;
    for (int kk = 0; kk < a.k; ++kk) sumy += a.colsum_gw[kk] * a.hstat_in[ESPM_HS_ROWSUM + kk];
    if (a.compute_loss) a.hist_slot[ESPM_HI_KLX] = (double)a.xscale * 0.6931471805599453 * v[ESPM_HP_KL];
    a.hist_slot[ESPM_HI_REG] = v[ESPM_HP_REG];
    a.hist_slot[ESPM_HI_LAP] = v[ESPM_HP_LAP];
    a.hist_slot[ESPM_HI_SUMY] = sumy;
    a.hist_slot[ESPM_HI_BAD] = v[ESPM_HP_BAD];
    if (a.hstat_out) {
      for (int kk = 0; kk < KP; ++kk) {
        a.hstat_out[ESPM_HS_ROWSUM + kk] = v[ESPM_HP_ROWSUM + kk];
        a.hstat_out[ESPM_HS_MAX + kk] = v[4 + KP + kk];
      }
    }
  }
}

// ---- dispatch ---------------------------------------------------------------------------------
template <int K, typename XT, int PX, int NW>
static int launch_h(const HStepArgs& args, int nblk, hipStream_t stream) {
  const size_t lds = (size_t)NW * K * 64 * PX * sizeof(float);
  const size_t lds_min = (size_t)NW * (4 + 2 * K) * sizeof(double);
  const size_t bytes = lds > lds_min ? lds : lds_min;
  if (args.compute_loss)
    hipLaunchKernelGGL((h_step_kernel<K, XT, PX, NW, true>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  else
    hipLaunchKernelGGL((h_step_kernel<K, XT, PX, NW, false>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  return check_hip(hipGetLastError(), "h_step launch");
}

template <int K>
static int dispatch_h_k(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  if (x_dtype == ESPM_X_BF16) {
    if (tile_px == 512) return launch_h<K, bf16_t, 8, 4>(args, nblk, stream);
    if (tile_px == 128) return launch_h<K, bf16_t, 2, 16>(args, nblk, stream);
  } else {
    if (tile_px == 256) return launch_h<K, float, 4, 4>(args, nblk, stream);
    if (tile_px == 128) return launch_h<K, float, 2, 16>(args, nblk, stream);
  }
  return set_error(ESPM_EINVAL, "h_step: tile_px %d not available for x_dtype %d", tile_px, x_dtype);
}

int dispatch_h_step(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  switch (args.k) {
    case 1: return dispatch_h_k<1>(args, x_dtype, tile_px, nblk, stream);
    case 2: return dispatch_h_k<2>(args, x_dtype, tile_px, nblk, stream);
    case 3: return dispatch_h_k<3>(args, x_dtype, tile_px, nblk, stream);
    case 4: return dispatch_h_k<4>(args, x_dtype, tile_px, nblk, stream);
    case 5: return dispatch_h_k<5>(args, x_dtype, tile_px, nblk, stream);
    case 6: return dispatch_h_k<6>(args, x_dtype, tile_px, nblk, stream);
    case 7: return dispatch_h_k<7>(args, x_dtype, tile_px, nblk, stream);
    case 8: return dispatch_h_k<8>(args, x_dtype, tile_px, nblk, stream);
  }
  return set_error(ESPM_EUNSUPPORTED, "h_step: k=%d not built (1..%d)", args.k, ESPM_MAX_K);
}

int launch_h_finalize(const HFinalizeArgs& args, hipStream_t stream) {
  hipLaunchKernelGGL(h_finalize_kernel, dim3(1), dim3(256), 0, stream, args);
  return check_hip(hipGetLastError(), "h_finalize launch");
}

}  // namespace espm
