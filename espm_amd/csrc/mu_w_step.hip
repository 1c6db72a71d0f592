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
#include "mu_common.hpp"

namespace espm {

template <int K, typename XT, int CH>
__global__ __launch_bounds__(256) void w_accum_kernel(const WAccumArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = ((blockIdx.y * 4 + wave) * 64 + lane) * CH;
  const bool active = c0 < a.n_pad;
  constexpr int UP = 4;

  float gw[CH][K];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) gw[i][kk] = active ? a.gw_s[(size_t)(c0 + i) * KP + kk] : 1.f;

  float acc[CH][K];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[i][kk] = 0.f;

  const int j_begin = blockIdx.x * a.ppb;
  const int j_end = min(a.p, j_begin + a.ppb);
  const XT* xp = static_cast<const XT*>(a.x_pm) + (size_t)j_begin * a.n_pad + (active ? c0 : 0);
  const float* ht = a.h_t + (size_t)j_begin * KP;

  auto body = [&](const XVec<XT, CH>& xv, const float* hp) {
    float hk[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) hk[kk] = hp[kk];  // wave-uniform -> scalar loads
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

  int j = j_begin;
  for (; j + UP <= j_end; j += UP) {
    XVec<XT, CH> xv[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      if (active) xv[u].load(xp + (size_t)u * a.n_pad); else xv[u].zero();
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) body(xv[u], ht + u * KP);
    xp += (size_t)UP * a.n_pad;
    ht += UP * KP;
  }
  for (; j < j_end; ++j) {
    XVec<XT, CH> xv;
    if (active) xv.load(xp); else xv.zero();
    body(xv, ht);
    xp += a.n_pad;
    ht += KP;
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

__global__ __launch_bounds__(256) void w_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                        int nblk, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 4 <= nblk; b += 4) {
    s0 += slab[(size_t)b * total + e];
    s1 += slab[(size_t)(b + 1) * total + e];
    s2 += slab[(size_t)(b + 2) * total + e];
    s3 += slab[(size_t)(b + 3) * total + e];
  }
  for (; b < nblk; ++b) s0 += slab[(size_t)b * total + e];
  out[e] = (s0 + s1) + (s2 + s3);
}

// ---- W finish: one workgroup of 1024 threads --------------------------------------------------
constexpr int WF_THREADS = 1024;

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

__global__ __launch_bounds__(WF_THREADS) void w_finish_kernel(const WFinishArgs a) {
  __shared__ double scratch[(WF_THREADS / 64) * 2 * KP];
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
        for (int c = 0; c < a.n; ++c) s = fmaf(a.g[(size_t)c * a.m + mm], a.a[(size_t)kk * a.n_pad + c], s);
        gta = s;
      } else {
        gta = a.a[(size_t)kk * a.n_pad + mm];
      }
      numv[e] = a.w_old[e] * gta;
      denv[e] = (a.g ? a.colsum_g[mm] : 1.f) * (float)a.hstat[ESPM_HS_ROWSUM + kk];
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
    double sum_l = 0.0;
    for (int e = tid; e < M * k; e += WF_THREADS) {
      const int mm = e / k, kk = e - mm * k;
      float den = denv[e];
      if (a.simplex_w && (!a.simplex_rows || a.simplex_rows[mm])) den += (float)s_mid[kk];
      float wn = fmaxf(numv[e] / den, a.log_shift);
      if (a.fixed_w && a.fixed_w[e] >= 0.f) wn = a.fixed_w[e];
      a.w_new[e] = wn;
      sum_l += (double)wn;
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
    float4* dst = reinterpret_cast<float4*>(a.gw_s + (size_t)c * KP);
    dst[0] = make_float4(row[0], row[1], row[2], row[3]);
    dst[1] = make_float4(row[4], row[5], row[6], row[7]);
  }
  block_reduce<KP, KP>(cs, scratch);
  if (tid == 0)
    for (int kk = 0; kk < KP; ++kk) a.colsum_gw[kk] = cs[kk];
}

// ---- rel_H, base.py:324 ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rel_h_kernel(const float* __restrict__ h_old, const float* __restrict__ h_new,
                                                    const double* __restrict__ hstat_new, double* hist_slot, int k,
                                                    int p, int p_pad, double inv_count, float rel_tol) {
  __shared__ double scratch[4];
  double tot = 0.0;
  for (int kk = 0; kk < k; ++kk) tot += hstat_new[ESPM_HS_ROWSUM + kk];
  const float shift = (float)((double)rel_tol * tot * inv_count);
  float worst = 0.f;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < p; q += gridDim.x * blockDim.x) {
    for (int kk = 0; kk < k; ++kk) {
      const float hn = h_new[(size_t)kk * p_pad + q], ho = h_old[(size_t)kk * p_pad + q];
      worst = fmaxf(worst, fabsf(hn - ho) / (hn + shift));
    }
  }
  double v[1] = {(double)worst};
  block_reduce<1, 0>(v, scratch);
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(hist_slot + ESPM_HI_REL_H),
              (unsigned long long)__double_as_longlong(v[0]));
}

// ---- dispatch -----------------------------------------------------------------------------------
template <int K>
static int dispatch_w_k(const WAccumArgs& args, int x_dtype, int nblk, hipStream_t stream) {
  if (x_dtype == ESPM_X_BF16) {
    dim3 grid(nblk, (args.n_pad + 4 * 64 * 8 - 1) / (4 * 64 * 8));
    hipLaunchKernelGGL((w_accum_kernel<K, bf16_t, 8>), grid, dim3(256), 0, stream, args);
  } else {
    dim3 grid(nblk, (args.n_pad + 4 * 64 * 4 - 1) / (4 * 64 * 4));
    hipLaunchKernelGGL((w_accum_kernel<K, float, 4>), grid, dim3(256), 0, stream, args);
  }
  return check_hip(hipGetLastError(), "w_accum launch");
}

int dispatch_w_accum(const WAccumArgs& args, int k, int x_dtype, int nblk, hipStream_t stream) {
  switch (k) {
    case 1: return dispatch_w_k<1>(args, x_dtype, nblk, stream);
    case 2: return dispatch_w_k<2>(args, x_dtype, nblk, stream);
    case 3: return dispatch_w_k<3>(args, x_dtype, nblk, stream);
    case 4: return dispatch_w_k<4>(args, x_dtype, nblk, stream);
    case 5: return dispatch_w_k<5>(args, x_dtype, nblk, stream);
    case 6: return dispatch_w_k<6>(args, x_dtype, nblk, stream);
    case 7: return dispatch_w_k<7>(args, x_dtype, nblk, stream);
    case 8: return dispatch_w_k<8>(args, x_dtype, nblk, stream);
  }
  return set_error(ESPM_EUNSUPPORTED, "w_accum: k=%d not built (1..%d)", k, ESPM_MAX_K);
}

int launch_w_reduce(const float* slab, float* out, int nblk, int total, hipStream_t stream) {
  hipLaunchKernelGGL(w_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, slab, out, nblk, total);
  return check_hip(hipGetLastError(), "w_reduce launch");
}

int launch_w_finish(const WFinishArgs& args, hipStream_t stream) {
  hipLaunchKernelGGL(w_finish_kernel, dim3(1), dim3(WF_THREADS), 0, stream, args);
  return check_hip(hipGetLastError(), "w_finish launch");
}

int launch_rel_h(const float* h_old, const float* h_new, const double* hstat_new, double* hist_slot, int k, int p,
                 int p_pad, double inv_count, float rel_tol, hipStream_t stream) {
  int blocks = (p + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rel_h_kernel, dim3(blocks), dim3(256), 0, stream, h_old, h_new, hstat_new, hist_slot, k, p,
                     p_pad, inv_count, rel_tol);
  return check_hip(hipGetLastError(), "rel_h launch");
}

}  // namespace espm
