// Small kernels around the two streaming passes: layout packing of X, statistics of an H buffer,
// the stand-alone simplex root finder (module-level dichotomy_simplex) and the stand-alone
// Laplacian product H @ L.
#include "mu_common.hpp"

namespace espm {

// ---- X packing: src (n, p) or (p, n), f32/f64 -> channel-major (n, p_pad) and pixel-major (p, n_pad)
template <typename DT>
__device__ __forceinline__ DT to_store(float v);
template <>
__device__ __forceinline__ float to_store<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t to_store<bf16_t>(float v) {
  // round to nearest even; X is finite and non-negative on this path (base.py:519-528)
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <>
__device__ __forceinline__ uint8_t to_store<uint8_t>(float v) { return (uint8_t)(v + 0.5f); }  // exact integers

template <typename ST, typename DT>
__global__ __launch_bounds__(256) void pack_x_kernel(const ST* __restrict__ src, int src_layout, int64_t ld, int n,
                                                     int p, DT* __restrict__ x_cm, DT* __restrict__ x_pm, int n_pad,
                                                     int p_pad, int x_tile, int n_cm) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  // tile[cc][jj] = X[c0 + cc, j0 + jj]
  for (int r = ty; r < 32; r += 8) {
    float v = 0.f;
    if (src_layout == ESPM_LAYOUT_CM) {
      const int c = c0 + r, j = j0 + tx;
      if (c < n && j < p) v = (float)src[(int64_t)c * ld + j];
      tile[r][tx] = v;
    } else {
      const int j = j0 + r, c = c0 + tx;
      if (c < n && j < p) v = (float)src[(int64_t)j * ld + c];
      tile[tx][r] = v;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, j = j0 + tx;
    if (x_cm && c < n_cm && j < p_pad)  // tile-major: [pixel block][channel row, zero padded to n_cm][x_tile]
      x_cm[((size_t)(j / x_tile) * n_cm + c) * x_tile + (j % x_tile)] = to_store<DT>(tile[r][tx]);
    const int j2 = j0 + r, c2 = c0 + tx;
    if (j2 < p && c2 < n_pad) x_pm[(size_t)j2 * n_pad + c2] = to_store<DT>(tile[tx][r]);
  }
}

template <typename ST>
static int pack_dispatch(const void* src, int src_layout, int64_t ld, int n, int p, void* x_cm, void* x_pm,
                         int x_dtype, int n_pad, int p_pad, int x_tile, int n_cm, hipStream_t stream) {
  dim3 grid((p_pad + 31) / 32, (n_cm + 31) / 32);
  if (x_dtype == ESPM_X_U8)
    hipLaunchKernelGGL((pack_x_kernel<ST, uint8_t>), grid, dim3(256), 0, stream, static_cast<const ST*>(src),
                       src_layout, ld, n, p, static_cast<uint8_t*>(x_cm), static_cast<uint8_t*>(x_pm), n_pad, p_pad, x_tile, n_cm);
  else if (x_dtype == ESPM_X_BF16)
    hipLaunchKernelGGL((pack_x_kernel<ST, bf16_t>), grid, dim3(256), 0, stream, static_cast<const ST*>(src),
                       src_layout, ld, n, p, static_cast<bf16_t*>(x_cm), static_cast<bf16_t*>(x_pm), n_pad, p_pad, x_tile, n_cm);
  else
    hipLaunchKernelGGL((pack_x_kernel<ST, float>), grid, dim3(256), 0, stream, static_cast<const ST*>(src),
                       src_layout, ld, n, p, static_cast<float*>(x_cm), static_cast<float*>(x_pm), n_pad, p_pad, x_tile, n_cm);
  return check_hip(hipGetLastError(), "pack_x launch");
}

int launch_pack_x(const void* src, int src_dtype, int src_layout, int64_t ld, int n, int p, void* x_cm, void* x_pm,
                  int x_dtype, int n_pad, int p_pad, int x_tile, int n_cm, hipStream_t stream) {
  if (src_dtype == ESPM_SRC_F64)
    return pack_dispatch<double>(src, src_layout, ld, n, p, x_cm, x_pm, x_dtype, n_pad, p_pad, x_tile, n_cm, stream);
  return pack_dispatch<float>(src, src_layout, ld, n, p, x_cm, x_pm, x_dtype, n_pad, p_pad, x_tile, n_cm, stream);
}

// ---- statistics of an H buffer (a workgroup per component; used at initialisation only) ----------
// (One workgroup for all components took 0.4 ms at the headline size, 2 ms with 17 components, 3.8 ms at configuration 5's: a single CU's
//  load rate.  The sums of a component are formed by the same threads in the same order as before.)
__global__ __launch_bounds__(1024) void hstat_kernel(const float* __restrict__ h, int k, int p, int p_pad,
                                                     double* __restrict__ out) {
  __shared__ double scratch[17 * 2];
  {
    const int kk = blockIdx.x;
    double v[2] = {0.0, 0.0};
    if (kk < k) {
      for (int q = threadIdx.x; q < p; q += blockDim.x) {
        const double x = h[(size_t)kk * p_pad + q];
        v[0] += x;
        v[1] = fmax(v[1], x);
      }
    }
    block_reduce<2, 1>(v, scratch);
    if (threadIdx.x == 0) {
      out[ESPM_HS_ROWSUM + kk] = v[0];
      out[ESPM_HS_MAX + kk] = v[1];
    }
  }
}

int launch_hstat(const float* h, int k, int p, int p_pad, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(hstat_kernel, dim3(KP), dim3(1024), 0, stream, h, k, p, p_pad, out);
  return check_hip(hipGetLastError(), "hstat launch");
}

// ---- stand-alone simplex multiplier (fp64) ------------------------------------------------------
constexpr int DS_MAXK = 64;

__global__ __launch_bounds__(256) void dichotomy_kernel(const double* __restrict__ num, const double* __restrict__ den,
                                                        int k, int p, int den_cols, double eps, double tol, int maxit,
                                                        double* __restrict__ nu_out, int32_t* __restrict__ status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p) return;
  // run-time-k variant of simplex_root (mu_common.hpp): shifted unknown delta = nu + d*
  double nmax = 0, dmin_all = INFINITY, dstar = INFINITY, nsum = 0;
  bool ok = true;
  for (int i = 0; i < k; ++i) {
    const double nn = num[(size_t)i * p + j], dd = den[(size_t)i * den_cols + (den_cols > 1 ? j : 0)];
    ok = ok && nn >= 0 && dd >= 0;
    if (nn > 0) dstar = fmin(dstar, dd);
    nmax = fmax(nmax, nn);
    dmin_all = fmin(dmin_all, dd);
    nsum += nn;
  }
  if (!ok || !(nsum > 0)) {
    atomicAdd(status, 1);
    nu_out[j] = NAN;
    return;
  }
  double lo = 0;
  for (int i = 0; i < k; ++i) {
    const double nn = num[(size_t)i * p + j], dd = den[(size_t)i * den_cols + (den_cols > 1 ? j : 0)];
    if (nn > 0) lo = fmax(lo, nn / 2 - (dd - dstar));
  }
  double hi = 2.0 * k * nmax + (dstar - dmin_all), x = lo, dxold = hi - lo;   // (this order: tiny numerators next to large denominators)
  for (int it = 0; it < maxit; ++it) {
    double f = -1, fp = 0;
    for (int i = 0; i < k; ++i) {
      const double nn = num[(size_t)i * p + j], dd = den[(size_t)i * den_cols + (den_cols > 1 ? j : 0)];
      const double inv = 1.0 / (x + (dd - dstar));
      const double t = nn > 0 ? nn * inv : 0.0;
      if (t > eps) {
        f += t;
        fp -= t * inv;
      } else {
        f += eps;
      }
    }
    if (fabs(f) <= tol) break;
    if (f > 0) lo = x; else hi = x;
    double dx = fp < 0 ? -f / fp : 0.0;
    double xn = x + dx;
    if (!(fp < 0) || !(xn > lo && xn < hi) || fabs(dx) > 0.5 * fabs(dxold)) {
      dx = (hi - lo) / 2;
      xn = lo + dx;
    }
    dxold = dx;
    if (xn == x) break;
    x = xn;
  }
  x -= dstar;
  nu_out[j] = x;
}

int launch_dichotomy(const double* num, const double* den, int k, int p, int den_cols, double eps, double tol,
                     int maxit, double* nu_out, int32_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(dichotomy_kernel, dim3((p + 255) / 256), dim3(256), 0, stream, num, den, k, p, den_cols, eps,
                     tol, maxit, nu_out, status);
  return check_hip(hipGetLastError(), "dichotomy launch");
}

// ---- the H update's per-pixel root as a launch of its own (fp32, simplex_root<float, K> exactly as h_epilogue inlines it) ----
// One thread per column: delta (the shifted unknown, nu = delta - min{den_i : num_i > 0}) and the shifted denominators e (k, p), so that a
// caller can evaluate f = sum max(num / (delta + e), eps) - 1 in higher precision: what the tests hold the routine to.
template <int K>
__global__ __launch_bounds__(256) void simplex_root_f32_kernel(const float* __restrict__ num, const float* __restrict__ den, int p, float eps, float tol,
                                                               int maxit, int fast_exit, float* __restrict__ delta_out, float* __restrict__ e_out,
                                                               int32_t* __restrict__ status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p) return;
  float nv[K], dv[K], e[K], delta;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    nv[i] = num[(size_t)i * p + j];
    dv[i] = den[(size_t)i * p + j];
  }
  if (!simplex_root<float, K>(nv, dv, K, eps, tol, maxit, delta, e, fast_exit != 0)) atomicAdd(status, 1);
  delta_out[j] = delta;
#pragma unroll
  for (int i = 0; i < K; ++i) e_out[(size_t)i * p + j] = e[i];
}

int launch_simplex_root_f32(const float* num, const float* den, int k, int p, float eps, float tol, int maxit, int fast_exit, float* delta_out,
                            float* e_out, int32_t* status, hipStream_t stream) {
  switch (k) {
#define ESPM_X(KK) \
  case KK: hipLaunchKernelGGL(simplex_root_f32_kernel<KK>, dim3((p + 255) / 256), dim3(256), 0, stream, num, den, p, eps, tol, maxit, fast_exit, delta_out, e_out, status); break;
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
    default: return set_error(ESPM_EUNSUPPORTED, "simplex_root_f32: k=%d not built (%d..%d)", k, ESPM_MIN_K, ESPM_MAX_K);
  }
  return check_hip(hipGetLastError(), "simplex_root_f32 launch");
}

// ---- the other two multipliers of espm/estimators/dicotomy.py as module-level functions (fp64, one thread per column) ----
// acc (dicotomy.py:57-82):  sum_k max(sqrt((b_kj + nu)^2 + 4 a c_kj) - nu - b_kj, 2 a eps) = 2 a
// pg  (dicotomy.py:84-108): sum_k max(a_kj + nu, eps) = 1
// Both sums are monotone in nu; the reference's bracket, a Newton iteration kept inside it, per-column convergence.
__global__ __launch_bounds__(256) void dichotomy_acc_kernel(double a, const double* __restrict__ b, const double* __restrict__ c,
                                                            int k, int p, int b_cols, double eps, double tol, int maxit,
                                                            double* __restrict__ nu_out, int32_t* __restrict__ status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p) return;
  auto bb = [&](int i) { return b[(size_t)i * b_cols + (b_cols > 1 ? j : 0)]; };
  double bmax = -INFINITY, bsum = 0.0;
  bool ok = a > 0;
  for (int i = 0; i < k; ++i) {
    const double cc = c[(size_t)i * p + j];
    ok = ok && cc >= 0;
    bmax = fmax(bmax, bb(i) * bb(i) / a + 2 * a + 2 * (bb(i) + cc));
    bsum += bb(i);
  }
  if (!ok) {
    atomicAdd(status, 1);
    nu_out[j] = NAN;
    return;
  }
  double hi = k * bmax * 1.5 + 1e-3, lo = -(2 * a + bsum) / k * 1.1 - 1e-3;
  const double floor_g = 2 * a * eps;
  double x = fmin(fmax(0.0, lo), hi), dxold = hi - lo;
  for (int it = 0; it < maxit; ++it) {
    double f = -2 * a, fp = 0.0;
    for (int i = 0; i < k; ++i) {
      const double s = bb(i) + x, q = 4 * a * c[(size_t)i * p + j];
      const double r = sqrt(s * s + q);
      const double g = s >= 0 ? q / (r + s) : r - s;
      if (g > floor_g) {
        f += g;
        fp -= g / r;
      } else {
        f += floor_g;
      }
    }
    if (fabs(f) <= tol) break;
    if (f > 0) lo = x; else hi = x;
    double dx = fp < 0 ? -f / fp : 0.0;
    double xn = x + dx;
    if (!(fp < 0) || !(xn > lo && xn < hi) || fabs(dx) > 0.5 * fabs(dxold)) {
      dx = (hi - lo) / 2;
      xn = lo + dx;
    }
    dxold = dx;
    if (xn == x) break;
    x = xn;
  }
  nu_out[j] = x;
}

__global__ __launch_bounds__(256) void dichotomy_pg_kernel(const double* __restrict__ a, int k, int p, double eps, double tol,
                                                           int maxit, double* __restrict__ nu_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p) return;
  double amin = INFINITY;
  for (int i = 0; i < k; ++i) amin = fmin(amin, a[(size_t)i * p + j]);
  double x = 1.0 / k - amin;   // right end of the bracket: every term is active there and the sum is >= 1
  for (int it = 0; it < maxit; ++it) {
    double f = -1.0, cnt = 0.0;
    for (int i = 0; i < k; ++i) {
      const double t = a[(size_t)i * p + j] + x;
      if (t > eps) { f += t; cnt += 1.0; } else { f += eps; }
    }
    if (fabs(f) <= tol || !(cnt > 0)) break;
    x -= f / cnt;   // convex, piecewise linear, increasing: Newton from the right is monotone and finite
  }
  nu_out[j] = x;
}

int launch_dichotomy_acc(double a, const double* b, const double* c, int k, int p, int b_cols, double eps, double tol, int maxit,
                         double* nu_out, int32_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(dichotomy_acc_kernel, dim3((p + 255) / 256), dim3(256), 0, stream, a, b, c, k, p, b_cols, eps, tol, maxit, nu_out,
                     status);
  return check_hip(hipGetLastError(), "dichotomy_acc launch");
}
int launch_dichotomy_pg(const double* a, int k, int p, double eps, double tol, int maxit, double* nu_out, hipStream_t stream) {
  hipLaunchKernelGGL(dichotomy_pg_kernel, dim3((p + 255) / 256), dim3(256), 0, stream, a, k, p, eps, tol, maxit, nu_out);
  return check_hip(hipGetLastError(), "dichotomy_pg launch");
}

// ---- stand-alone H @ L ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void laplacian_kernel(const float* __restrict__ h, int k, int nx, int ny, int64_t ld,
                                                        float* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nx * ny) return;
  for (int kk = 0; kk < k; ++kk) {
    const float* row = h + (size_t)kk * ld;
    out[(size_t)kk * ld + q] = stencil_hl(row, nullptr, nullptr, q, nx, ny, row[q]);
  }
}

int launch_laplacian(const float* h, int k, int nx, int ny, int64_t ld, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(laplacian_kernel, dim3((nx * ny + 255) / 256), dim3(256), 0, stream, h, k, nx, ny, ld, out);
  return check_hip(hipGetLastError(), "laplacian launch");
}


// ---- linesearch on the Laplacian surrogate (espm/estimators/surrogates.py:65-149, smooth_nmf.py:376-381) -------
// d = g(H, Ht) - lambda/2 tr(H L H^T) needs, over all pixels, with Ht = the H before the update and H = the new one:
//   t1 = sum Ht (Ht L), t2 = sum (Ht L) H, b = sum H (H L), sq = sum (Ht - H)^2 (the quadratic surrogate's term) and per
// component dg_k = sum_j Ht log(Ht / H) - Ht + H (the caller weights dg_k with max_j H_kj, which the H-step already
// leaves in hstat).  One thread per pixel, stencil
// as in the H-step; workgroup partials (field-major, LS_FIELDS rows) then a one-workgroup sum in fixed order.
constexpr int LS_FIELDS = 4 + KP;
// Sharded image: the image rows above / below the local block come from the neighbours' boundary rows, (k, ny) each - of
// the OLD H (old_top / old_bot) and of the new one (new_top / new_bot); null at the image edge and on one GPU.
__global__ __launch_bounds__(256) void linesearch_terms_kernel(const float* __restrict__ h_old, const float* __restrict__ h_new,
                                                               int k, int p, int p_pad, int nx, int ny, int grid_mode,
                                                               const float* __restrict__ old_top, const float* __restrict__ old_bot,
                                                               const float* __restrict__ new_top, const float* __restrict__ new_bot,
                                                               double* __restrict__ part) {
  __shared__ double scratch[5 * LS_FIELDS];
  double v[LS_FIELDS];
#pragma unroll
  for (int i = 0; i < LS_FIELDS; ++i) v[i] = 0.0;
  for (int q = blockIdx.x * 512 + threadIdx.x; q < min(p, (int)(blockIdx.x + 1) * 512); q += 256) {
    for (int kk = 0; kk < k; ++kk) {
      const float* ro = h_old + (size_t)kk * p_pad;
      const float* rn = h_new + (size_t)kk * p_pad;
      const float ho = ro[q], hn = rn[q];
      const size_t hk = (size_t)kk * ny;
      const float lo = grid_mode ? stencil_hl(ro, old_top ? old_top + hk : nullptr, old_bot ? old_bot + hk : nullptr, q, nx, ny, ho)
                                 : ho;   // L = identity without a grid (base.py:289-291)
      const float ln = grid_mode ? stencil_hl(rn, new_top ? new_top + hk : nullptr, new_bot ? new_bot + hk : nullptr, q, nx, ny, hn) : hn;
      v[0] += (double)ho * (double)lo;
      v[1] += (double)lo * (double)hn;
      v[2] += (double)hn * (double)ln;
      v[3] += ((double)ho - (double)hn) * ((double)ho - (double)hn);
      v[4 + kk] += (double)ho * log((double)ho / (double)hn) - (double)ho + (double)hn;
    }
  }
  block_reduce<LS_FIELDS, LS_FIELDS>(v, scratch);
  if (threadIdx.x == 0)
    for (int i = 0; i < LS_FIELDS; ++i) part[(size_t)i * gridDim.x + blockIdx.x] = v[i];
}
__global__ __launch_bounds__(256) void linesearch_sum_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  __shared__ double scratch[5 * LS_FIELDS];
  double v[LS_FIELDS];
#pragma unroll
  for (int i = 0; i < LS_FIELDS; ++i) {
    v[i] = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) v[i] += part[(size_t)i * nblk + b];
  }
  block_reduce<LS_FIELDS, LS_FIELDS>(v, scratch);
  if (threadIdx.x == 0)
    for (int i = 0; i < LS_FIELDS; ++i) out[i] = v[i];
}

int launch_linesearch_terms(const float* h_old, const float* h_new, int k, int p, int p_pad, int nx, int ny, int grid_mode,
                            const float* old_top, const float* old_bot, const float* new_top, const float* new_bot,
                            double* part, double* out, hipStream_t stream) {
  const int nblk = (p + 511) / 512;
  hipLaunchKernelGGL(linesearch_terms_kernel, dim3(nblk), dim3(256), 0, stream, h_old, h_new, k, p, p_pad, nx, ny, grid_mode, old_top,
                     old_bot, new_top, new_bot, part);
  hipLaunchKernelGGL(linesearch_sum_kernel, dim3(1), dim3(256), 0, stream, part, nblk, out);
  return check_hip(hipGetLastError(), "linesearch_terms launch");
}


// ---- sharded image: per-rank exchange record ------------------------------------------------------
// One record per rank and iteration:  [ A (k*n_pad f32) | hstat of the new H
// (ESPM_HS_STRIDE f64) | first owned image row of the new H (k*ny f32) | last owned row (k*ny f32) ].
// The A block starts the record, the f64 block is 8-byte aligned because k*n_pad is a multiple of 8.
__global__ __launch_bounds__(256) void shard_pack_kernel(const float* __restrict__ a, const double* __restrict__ hstat,
                                                         const float* __restrict__ h_new, int k, int n_pad, int nx,
                                                         int ny, int p_pad, int with_halo, unsigned char* rec) {
  const int na = k * n_pad;
  float* ra = reinterpret_cast<float*>(rec);
  double* rs = reinterpret_cast<double*>(rec + (size_t)na * 4);
  float* rt = reinterpret_cast<float*>(rec + (size_t)na * 4 + ESPM_HS_STRIDE * 8);
  float* rb = rt + (size_t)k * ny;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int e = tid; e < na; e += nt) ra[e] = a[e];
  for (int e = tid; e < ESPM_HS_STRIDE; e += nt) rs[e] = hstat[e];
  if (with_halo) {
    for (int e = tid; e < k * ny; e += nt) {
      const int kk = e / ny, j = e - kk * ny;
      rt[e] = h_new[(size_t)kk * p_pad + j];
      rb[e] = h_new[(size_t)kk * p_pad + (size_t)(nx - 1) * ny + j];
    }
  }
}

// Fixed-order sum over ranks (bit-identical on every rank), global row sums / maxima.
// With bparts (the simplex over W with G = identity, mu_w_step.hip: w_simplex_update_kernel; n_pad a multiple of 32): every
// half wave also leaves sum, maximum and count of the positive numerators W A of its 32 entries in bparts[3 * (entry / 32) ..],
// as w_reduce_kernel does for one GPU.
__global__ __launch_bounds__(256) void shard_combine_kernel(const unsigned char* __restrict__ recs, int world,
                                                            size_t stride, int na, float* __restrict__ a_out,
                                                            double* __restrict__ hstat_out, const float* __restrict__ bw_old,
                                                            double* __restrict__ bparts, int bn, int bk, int bn_pad) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int e0 = tid & ~31; e0 < na; e0 += nt) {   // (whole half waves enter every trip)
    const int e = e0 + (tid & 31);
    float s = 0.f;
    if (e < na) {
      for (int r = 0; r < world; ++r) s += reinterpret_cast<const float*>(recs + r * stride)[e];
      a_out[e] = s;
    }
    if (bparts) {
      const int kk = e / bn_pad, c = e - kk * bn_pad;
      const float num = (e < na && c < bn) ? bw_old[(size_t)c * bk + kk] * s : 0.f;
      double sum = num > 0.f ? (double)num : 0.0, cnt = num > 0.f ? 1.0 : 0.0;
      float mx = fmaxf(num, 0.f);
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      }
      if ((tid & 31) == 0) {
        bparts[3 * (size_t)(e0 / 32)] = sum;
        bparts[3 * (size_t)(e0 / 32) + 1] = (double)mx;
        bparts[3 * (size_t)(e0 / 32) + 2] = cnt;
      }
    }
  }
  for (int e = tid; e < ESPM_HS_STRIDE; e += nt) {
    double s = 0.0;
    for (int r = 0; r < world; ++r) {
      const double v = reinterpret_cast<const double*>(recs + r * stride + (size_t)na * 4)[e];
      s = e < ESPM_HS_MAX ? s + v : fmax(s, v);
    }
    hstat_out[e] = s;
  }
}

int launch_shard_pack(const float* a, const double* hstat, const float* h_new, int k, int n_pad, int nx, int ny,
                      int p_pad, int with_halo, void* rec, hipStream_t stream) {
  int blocks = (k * n_pad + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(shard_pack_kernel, dim3(blocks), dim3(256), 0, stream, a, hstat, h_new, k, n_pad, nx, ny, p_pad,
                     with_halo, static_cast<unsigned char*>(rec));
  return check_hip(hipGetLastError(), "shard_pack launch");
}

int launch_shard_combine(const void* recs, int world, size_t stride, int na, float* a_out, double* hstat_out,
                         hipStream_t stream, const float* bw_old, double* bparts, int n, int k, int n_pad) {
  int blocks = (na + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(shard_combine_kernel, dim3(blocks), dim3(256), 0, stream,
                     static_cast<const unsigned char*>(recs), world, stride, na, a_out, hstat_out, bw_old, bparts, n, k, n_pad);
  return check_hip(hipGetLastError(), "shard_combine launch");
}

}  // namespace espm
