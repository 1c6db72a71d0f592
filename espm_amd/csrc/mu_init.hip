// Device side of the initialisation (init on device; espm_amd/init_device.py is the host side).
//
// espm_lu_pl: the LU normaliser of scikit-learn's randomized range finder, which the reference's NNDSVD initialisation runs 14 times
// (espm/estimators/updates.py:179 -> sklearn.decomposition._nmf._initialize_nmf -> sklearn.utils.extmath._randomized_range_finder,
// power_iteration_normalizer="LU": Q, _ = scipy.linalg.lu(A, permute_l=True)) on tall matrices of n_components + 10 columns.
// As torch operations it is ~8 launches per column, 1.35 ms per call at ANY height (launch latency: the 2048 x 15 matrix costs what
// the 262144 x 15 one does) - 19 ms of a 160 ms fit.  Here: one launch per column.  The launch of column j finds the pivot from the
// per-workgroup candidates the launch before left (largest modulus among the rows that have not been a pivot, the first such row on
// ties: LAPACK's getrf), forms the multipliers of its rows, updates the remaining columns of its rows and leaves its candidate for
// column j + 1.  Row i of the result P L: the multipliers of the steps before row i became a pivot, 1 at its own step, 0 after.
// Arithmetic as the torch formulation it replaces (espm_amd/init_device._lu_pl: quotient, product and difference rounded one by one).
#include "mu_common.hpp"

namespace espm {

struct LuCand {
  double val;        // modulus of the candidate (-1: no row of this workgroup is free)
  long long row;
};

template <typename T>
struct LuArgs {
  const T* a;        // (m, r), leading dimension ld: read by the launch of column -1 only
  T* work;           // (m, r) contiguous: the matrix being eliminated
  T* out;            // (m, r) contiguous: P L
  unsigned char* free_rows;
  const LuCand* cand_in;
  LuCand* cand_out;
  long long ld;
  int m, r, j, ncand, rows_per_wg;
};

__device__ __forceinline__ bool lu_better(double v, long long row, double bv, long long brow) { return v > bv || (v == bv && row < brow); }

// the best candidate of the block in every thread (256 threads)
__device__ __forceinline__ LuCand lu_block_best(double v, long long row) {
  __shared__ double s_v[4];
  __shared__ long long s_r[4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(v, off, 64);
    const long long orow = __shfl_xor(row, off, 64);
    if (lu_better(ov, orow, v, row)) { v = ov; row = orow; }
  }
  __syncthreads();   // (the arrays may still be read from the call before)
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = v; s_r[threadIdx.x >> 6] = row; }
  __syncthreads();
  LuCand b{s_v[0], s_r[0]};
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (lu_better(s_v[w], s_r[w], b.val, b.row)) { b.val = s_v[w]; b.row = s_r[w]; }
  return b;
}

template <typename T>
__device__ __forceinline__ T lu_mul(T a, T b);
template <>
__device__ __forceinline__ float lu_mul<float>(float a, float b) { return __fmul_rn(a, b); }
template <>
__device__ __forceinline__ double lu_mul<double>(double a, double b) { return __dmul_rn(a, b); }
template <typename T>
__device__ __forceinline__ T lu_sub(T a, T b);
template <>
__device__ __forceinline__ float lu_sub<float>(float a, float b) { return __fsub_rn(a, b); }
template <>
__device__ __forceinline__ double lu_sub<double>(double a, double b) { return __dsub_rn(a, b); }

// j = -1: copy A into the work matrix, every row free, candidates of column 0.  j >= 0: eliminate column j.
template <typename T, int RMAX>
__global__ __launch_bounds__(256) void lu_pl_kernel(const LuArgs<T> x) {
  __shared__ T s_prow[RMAX];
  __shared__ long long s_piv;
  __shared__ T s_pv;
  const int r = x.r, j = x.j;
  const long long row0 = (long long)blockIdx.x * x.rows_per_wg;
  const long long row1 = min((long long)x.m, row0 + x.rows_per_wg);
  if (j >= 0) {
    // the pivot: best of the candidates of the launch before
    double v = -2.0;
    long long row = 0x7fffffffffffffffLL;
    for (int c = threadIdx.x; c < x.ncand; c += 256) {
      const LuCand k = x.cand_in[c];
      if (lu_better(k.val, k.row, v, row)) { v = k.val; row = k.row; }
    }
    const LuCand best = lu_block_best(v, row);
    if (threadIdx.x == 0) s_piv = best.row;
    __syncthreads();
    const long long piv = s_piv;
    if ((int)threadIdx.x < r) s_prow[threadIdx.x] = x.work[piv * r + threadIdx.x];   // (the pivot row is not updated by this launch)
    __syncthreads();
    if (threadIdx.x == 0) s_pv = s_prow[j];
    __syncthreads();
  }
  double cv = -1.0;
  long long crow = 0x7fffffffffffffffLL;
  for (long long row = row0 + threadIdx.x; row < row1; row += 256) {
    T* w = x.work + row * r;
    bool fr;
    if (j < 0) {
      for (int c = 0; c < r; ++c) w[c] = x.a[row * x.ld + c];
      x.free_rows[row] = 1;
      fr = true;
    } else {
      const long long piv = s_piv;
      const T pv = s_pv;
      fr = x.free_rows[row] != 0;
      const T col = w[j];
      T mult = (fr && pv != T(0)) ? col / pv : T(0);   // (a zero pivot: the column is zero in the free rows, LAPACK leaves zeros)
      if (row == piv) {
        mult = T(1);
        x.free_rows[row] = 0;
        fr = false;
      }
      x.out[row * r + j] = mult;
      if (fr)
        for (int c = j + 1; c < r; ++c) w[c] = lu_sub<T>(w[c], lu_mul<T>(mult, s_prow[c]));
    }
    if (fr && j + 1 < r) {
      const double v = fabs((double)w[j + 1]);
      if (lu_better(v, row, cv, crow)) { cv = v; crow = row; }
    }
  }
  if (j + 1 < r) {
    const LuCand best = lu_block_best(cv, crow);
    if (threadIdx.x == 0) x.cand_out[blockIdx.x] = best;
  }
}

static int lu_grid(int m, int* rows_per_wg) {
  int nwg = (m + 255) / 256;
  if (nwg > 1024) nwg = 1024;
  if (nwg < 1) nwg = 1;
  *rows_per_wg = (m + nwg - 1) / nwg;
  return (m + *rows_per_wg - 1) / *rows_per_wg;
}

size_t lu_pl_scratch_bytes(int m, int r, int dtype) {
  const size_t el = dtype == ESPM_SRC_F64 ? 8 : 4;
  int rpw;
  const int nwg = lu_grid(m, &rpw);
  size_t b = ((size_t)m * r * el + 255) / 256 * 256;       // work matrix
  b += 2 * (((size_t)nwg * sizeof(LuCand) + 255) / 256 * 256);   // candidates, two generations
  b += ((size_t)m + 255) / 256 * 256;                       // free rows
  return b;
}

template <typename T>
static int lu_pl_run(const void* a, int m, int r, long long ld, void* out, unsigned char* scratch, hipStream_t stream) {
  constexpr int RMAX = 64;
  int rpw;
  const int nwg = lu_grid(m, &rpw);
  LuArgs<T> x;
  x.a = static_cast<const T*>(a);
  x.work = reinterpret_cast<T*>(scratch);
  size_t off = ((size_t)m * r * sizeof(T) + 255) / 256 * 256;
  LuCand* cand[2];
  cand[0] = reinterpret_cast<LuCand*>(scratch + off);
  off += ((size_t)nwg * sizeof(LuCand) + 255) / 256 * 256;
  cand[1] = reinterpret_cast<LuCand*>(scratch + off);
  off += ((size_t)nwg * sizeof(LuCand) + 255) / 256 * 256;
  x.free_rows = scratch + off;
  x.out = static_cast<T*>(out);
  x.ld = ld;
  x.m = m;
  x.r = r;
  x.ncand = nwg;
  x.rows_per_wg = rpw;
  for (int j = -1; j < r; ++j) {
    x.j = j;
    x.cand_in = cand[(j + 2) & 1];      // written by the launch of column j - 1 ...
    x.cand_out = cand[(j + 1) & 1];     // ... while this one writes the other generation
    hipLaunchKernelGGL((lu_pl_kernel<T, RMAX>), dim3(nwg), dim3(256), 0, stream, x);
  }
  return check_hip(hipGetLastError(), "lu_pl launch");
}

int launch_lu_pl(const void* a, int dtype, int m, int r, long long ld, void* out, void* scratch, hipStream_t stream) {
  if (dtype == ESPM_SRC_F64) return lu_pl_run<double>(a, m, r, ld, out, static_cast<unsigned char*>(scratch), stream);
  return lu_pl_run<float>(a, m, r, ld, out, static_cast<unsigned char*>(scratch), stream);
}

}  // namespace espm

extern "C" {

size_t espm_lu_pl_scratch_bytes(int m, int r, int dtype) {
  if (m < 1 || r < 1 || r > 64 || (dtype != ESPM_SRC_F32 && dtype != ESPM_SRC_F64)) return 0;
  return espm::lu_pl_scratch_bytes(m, r, dtype);
}

int espm_lu_pl(const void* a, int dtype, int m, int r, int64_t ld, void* out, void* scratch, size_t scratch_bytes, espm_stream_t stream) {
  ESPM_REQUIRE(a && out && scratch, "lu_pl: NULL pointer");
  ESPM_REQUIRE(dtype == ESPM_SRC_F32 || dtype == ESPM_SRC_F64, "lu_pl: dtype %d is neither ESPM_SRC_F32 nor ESPM_SRC_F64", dtype);
  ESPM_REQUIRE(r >= 1 && r <= 64 && m >= r && ld >= r, "lu_pl: a tall (m, r) matrix with r <= 64 columns is expected, got m=%d r=%d ld=%lld", m, r, (long long)ld);
  ESPM_REQUIRE(scratch_bytes >= espm::lu_pl_scratch_bytes(m, r, dtype), "lu_pl: scratch of %zu bytes, %zu are needed (espm_lu_pl_scratch_bytes)",
               scratch_bytes, espm::lu_pl_scratch_bytes(m, r, dtype));
  return espm::launch_lu_pl(a, dtype, m, r, (long long)ld, out, scratch, static_cast<hipStream_t>(stream));
}

}  // extern "C"
