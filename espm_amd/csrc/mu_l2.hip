// Frobenius ("l2") branch of the multiplicative updates: espm/estimators/updates.py:31-36 (W) and :109-118 (H),
// reachable in the reference only through direct calls of the step functions (l2=True is reset by SmoothNMF unless
// algo="l2_surrogate", smooth_nmf.py:233-237; espm/tests/test_updates.py:457-568 exercise it).
//
//   H' = max(H * (GW^T X) / ((GW^T GW) H + nu), eps)          num = GW^T X is the streaming contraction over the
//                                                             channels: the dense H-step kernel with R = X (mu_h_kernel.hpp,
//                                                             template flag L2), the k x k Gram matrix goes to the epilogue
//   W' = max(W * (G^T (X H^T)) / (G^T G W H H^T), eps)        A = X H^T is the streaming contraction over the pixels:
//                                                             the dense W accumulation with R = X, then one workgroup
// The four products of the Frobenius rule - GW^T X, GW^T GW, X H^T, H H^T - are k-column contractions (k <= 8): 2k
// flops per byte of X, three orders of magnitude below the matrix cores' balance, so they ride in the same VALU
// streaming kernels as the KL rule (DESIGN.md section 4 has the MFMA measurement for Y = GW H).
#include "mu_common.hpp"

namespace espm {

// out[a][b] = sum_r M[r][a] M[r][b] for M (rows, KP) fp32, a, b < k.  Workgroup = GRAM_GROUPS row groups x KP * KP
// (a, b) pairs; partials [KP * KP][nblk] doubles, then one workgroup sums them in fixed order.
constexpr int GRAM_PAIRS = KP * KP, GRAM_GROUPS = GRAM_PAIRS >= 256 ? 1 : 256 / GRAM_PAIRS;   // 64 x 4, 256 x 1 in the wide build, 1024 x 1 with KP = 32
constexpr int GRAM_THREADS = GRAM_PAIRS * GRAM_GROUPS;
__global__ __launch_bounds__(GRAM_THREADS) void gram_partial_kernel(const float* __restrict__ m, int rows, int k, double* __restrict__ part) {
  __shared__ double s[GRAM_GROUPS][GRAM_PAIRS];
  const int e = threadIdx.x % GRAM_PAIRS, grp = threadIdx.x / GRAM_PAIRS;
  const int ia = e / KP, ib = e % KP;
  double acc = 0.0;
  if (ia < k && ib < k) {
    for (int r = blockIdx.x * GRAM_GROUPS + grp; r < rows; r += gridDim.x * GRAM_GROUPS)
      acc += (double)m[(size_t)r * KP + ia] * (double)m[(size_t)r * KP + ib];
  }
  s[grp][e] = acc;
  __syncthreads();
  if (grp == 0) {
    double t = s[0][e];
    if constexpr (GRAM_GROUPS == 4) t = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
    part[(size_t)e * gridDim.x + blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(GRAM_PAIRS) void gram_sum_kernel(const double* __restrict__ part, int nblk, float* __restrict__ out) {
  const int e = threadIdx.x;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += part[(size_t)e * nblk + b];
  out[e] = (float)t;
}

int launch_gram(const float* m, int rows, int k, double* part, int part_cap, float* out, hipStream_t stream) {
  int nblk = (rows + 1023) / 1024;
  const int cap = part_cap / (KP * KP);
  if (cap < 1) return set_error(ESPM_EINVAL, "gram: scratch of %d doubles is smaller than %d", part_cap, KP * KP);
  if (nblk > cap) nblk = cap;
  if (nblk > 256) nblk = 256;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nblk), dim3(GRAM_THREADS), 0, stream, m, rows, k, part);
  hipLaunchKernelGGL(gram_sum_kernel, dim3(1), dim3(GRAM_PAIRS), 0, stream, part, nblk, out);
  return check_hip(hipGetLastError(), "gram launch");
}

// W' = max(W / (G^T G W H H^T) * (G^T A), eps) with A = X H^T (k, n_pad), updates.py:31-36, :70-76; one workgroup.
__global__ __launch_bounds__(1024) void w_finish_l2_kernel(const float* __restrict__ a, int n, int n_pad, int m, int k,
                                                           const float* __restrict__ g, const float* __restrict__ gtg,
                                                           const float* __restrict__ hh, const float* __restrict__ w_old,
                                                           float* __restrict__ w_new, const float* __restrict__ fixed_w,
                                                           float log_shift) {
  const int M = m > 0 ? m : n;
  for (int o = threadIdx.x; o < M * k; o += blockDim.x) {
    const int mm = o / k, kk = o - mm * k;
    double gxh = 0.0;   // (G^T (X H^T))[mm][kk]
    if (m > 0) {
      for (int c = 0; c < n; ++c) gxh += (double)g[(size_t)c * m + mm] * (double)a[(size_t)kk * n_pad + c];
    } else {
      gxh = a[(size_t)kk * n_pad + mm];
    }
    double den = 0.0;   // (G^T G W H H^T)[mm][kk]
    for (int l = 0; l < k; ++l) {
      double ggw = 0.0;
      if (m > 0) {
        for (int m2 = 0; m2 < m; ++m2) ggw += (double)gtg[(size_t)mm * m + m2] * (double)w_old[(size_t)m2 * k + l];
      } else {
        ggw = w_old[(size_t)mm * k + l];
      }
      den += ggw * (double)hh[l * KP + kk];
    }
    float v = fmaxf((float)((double)w_old[o] / den * gxh), log_shift);
    if (fixed_w) {
      const float fx = fixed_w[o];
      if (fx >= 0.f) v = fx;
    }
    w_new[o] = v;
  }
}

int launch_w_finish_l2(const float* a, int n, int n_pad, int m, int k, const float* g, const float* gtg, const float* hh,
                       const float* w_old, float* w_new, const float* fixed_w, float log_shift, hipStream_t stream) {
  hipLaunchKernelGGL(w_finish_l2_kernel, dim3(1), dim3(1024), 0, stream, a, n, n_pad, m, k, g, gtg, hh, w_old, w_new, fixed_w,
                     log_shift);
  return check_hip(hipGetLastError(), "w_finish (l2) launch");
}

}  // namespace espm
