// Shared device helpers for libespm_mu (gfx950 / CDNA4 only: wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "espm_mu.h"

namespace espm {

constexpr int KP = ESPM_KP;
static_assert(KP == 8 || KP == 16 || KP == 32, "ESPM_KP: 8, 16 or 32");
static_assert(ESPM_MIN_K >= 1 && ESPM_MAX_K <= KP && ESPM_MIN_K <= ESPM_MAX_K, "ESPM_MIN_K .. ESPM_MAX_K must fit the stride");

// the component counts this build instantiates its kernels for: X(1) .. X(8), X(9) .. X(16) in the wide build, X(17) .. X(32) in the
// widest (KP = 32)
#if ESPM_MIN_K <= 8
#define ESPM_K_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#elif ESPM_MIN_K <= 16
#define ESPM_K_CASES(X) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#else
#define ESPM_K_CASES(X) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#endif

// one KP-strided row (gw_s, h_t) as 16-byte stores
__device__ __forceinline__ void store_row_kp(float* dst, const float (&row)[KP]) {
#pragma unroll
  for (int q = 0; q < KP / 4; ++q)
    reinterpret_cast<float4*>(dst)[q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
}
constexpr int WAVE = 64;

// Phase stamps of the fused kernel for tools/analysis/phase_clock.py: only in a library built with -DESPM_PHASE_CLOCK (never
// the product's build).  Thread 0 of a workgroup writes the 100 MHz wall clock into slot `id` of its 8 slots.
#ifdef ESPM_PHASE_CLOCK
static __device__ unsigned long long* espm_phase_buf = nullptr;
#define ESPM_PHASE_SLOTS 56   // 8 workgroup stamps, one per wave (16) for the end of the H walk and of the W walk, 16 more inside the phases (40..)
#define ESPM_PHASE_STAMP(id)                                                                                                 \
  do {                                                                                                                       \
    if (threadIdx.x == 0 && espm_phase_buf) espm_phase_buf[(size_t)blockIdx.x * ESPM_PHASE_SLOTS + (id)] = wall_clock64();   \
  } while (0)
// slot 43: where the workgroup ran - HW_ID (wave / SIMD / CU / SH / SE fields) in the low word, XCC_ID in the high one
#define ESPM_PHASE_WHERE()                                                                                                   \
  do {                                                                                                                       \
    if (threadIdx.x == 0 && espm_phase_buf) {                                                                                \
      unsigned hw, xcc;                                                                                                      \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                       \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                                     \
      espm_phase_buf[(size_t)blockIdx.x * ESPM_PHASE_SLOTS + 43] = ((unsigned long long)xcc << 32) | hw;                     \
    }                                                                                                                        \
  } while (0)
#define ESPM_WAVE_STAMP(base)                                                                                                              \
  do {                                                                                                                                     \
    if ((threadIdx.x & 63) == 0 && espm_phase_buf)                                                                                         \
      espm_phase_buf[(size_t)blockIdx.x * ESPM_PHASE_SLOTS + (base) + (threadIdx.x >> 6)] = wall_clock64();                                 \
  } while (0)
#else
#define ESPM_PHASE_STAMP(id) \
  do {                       \
  } while (0)
#define ESPM_WAVE_STAMP(base) \
  do {                        \
  } while (0)
#define ESPM_PHASE_WHERE() \
  do {                     \
  } while (0)
#endif

typedef uint16_t bf16_t;  // raw storage; converted with shifts (exact)
typedef float f2 __attribute__((ext_vector_type(2)));  // pairs of fp32: v_pk_fma_f32 / v_pk_mul_f32

int set_error(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define ESPM_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return espm::set_error(ESPM_EINVAL, __VA_ARGS__); \
  } while (0)

// ---- vector loads of PX consecutive X values as fp32 -------------------------------------
template <typename XT, int PX>
struct XVec;

template <>
struct XVec<bf16_t, 8> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float (&x)[8]) const {
    x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
    x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
    x[4] = __uint_as_float(v.z << 16); x[5] = __uint_as_float(v.z & 0xffff0000u);
    x[6] = __uint_as_float(v.w << 16); x[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  __device__ __forceinline__ void get2(f2 (&x)[4]) const {
    x[0] = f2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
    x[1] = f2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    x[2] = f2{__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u)};
    x[3] = f2{__uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
  }
  template <int T>
  __device__ __forceinline__ float elem() const {
    const uint32_t w = T < 2 ? v.x : (T < 4 ? v.y : (T < 6 ? v.z : v.w));
    return (T & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  }
};
template <>
struct XVec<bf16_t, 4> {
  uint2 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint2(0, 0); }
  __device__ __forceinline__ void get(float (&x)[4]) const {
    x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
    x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ __forceinline__ void get2(f2 (&x)[2]) const {
    x[0] = f2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
    x[1] = f2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
  }
};
template <>
struct XVec<bf16_t, 2> {
  uint32_t v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ void zero() { v = 0; }
  __device__ __forceinline__ void get(float (&x)[2]) const {
    x[0] = __uint_as_float(v << 16); x[1] = __uint_as_float(v & 0xffff0000u);
  }
  __device__ __forceinline__ void get2(f2 (&x)[1]) const {
    x[0] = f2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
  }
};
// 8-bit counts (lossless for integer data <= 255): v_cvt_f32_ubyteN widens straight from the packed dword
__device__ __forceinline__ float ub0(uint32_t u) { return (float)(u & 0xffu); }
__device__ __forceinline__ float ub1(uint32_t u) { return (float)((u >> 8) & 0xffu); }
__device__ __forceinline__ float ub2(uint32_t u) { return (float)((u >> 16) & 0xffu); }
__device__ __forceinline__ float ub3(uint32_t u) { return (float)(u >> 24); }
template <>
struct XVec<uint8_t, 8> {
  uint2 v;
  __device__ __forceinline__ void load(const uint8_t* p) { v = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint2(0, 0); }
  __device__ __forceinline__ void get2(f2 (&x)[4]) const {
    x[0] = f2{ub0(v.x), ub1(v.x)};
    x[1] = f2{ub2(v.x), ub3(v.x)};
    x[2] = f2{ub0(v.y), ub1(v.y)};
    x[3] = f2{ub2(v.y), ub3(v.y)};
  }
  template <int T>
  __device__ __forceinline__ float elem() const {
    const uint32_t w = T < 4 ? v.x : v.y;
    return (T & 3) == 0 ? ub0(w) : ((T & 3) == 1 ? ub1(w) : ((T & 3) == 2 ? ub2(w) : ub3(w)));
  }
};
template <>
struct XVec<uint8_t, 4> {
  uint32_t v;
  __device__ __forceinline__ void load(const uint8_t* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ void zero() { v = 0; }
  __device__ __forceinline__ void get2(f2 (&x)[2]) const {
    x[0] = f2{ub0(v), ub1(v)};
    x[1] = f2{ub2(v), ub3(v)};
  }
};
template <>
struct XVec<uint8_t, 2> {
  uint16_t v;
  __device__ __forceinline__ void load(const uint8_t* p) { v = *reinterpret_cast<const uint16_t*>(p); }
  __device__ __forceinline__ void zero() { v = 0; }
  __device__ __forceinline__ void get2(f2 (&x)[1]) const { x[0] = f2{ub0(v), ub1(v)}; }
};
template <>
struct XVec<float, 4> {
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void get(float (&x)[4]) const { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
  __device__ __forceinline__ void get2(f2 (&x)[2]) const { x[0] = f2{v.x, v.y}; x[1] = f2{v.z, v.w}; }
};
template <>
struct XVec<float, 2> {
  float2 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float2*>(p); }
  __device__ __forceinline__ void zero() { v = make_float2(0.f, 0.f); }
  __device__ __forceinline__ void get(float (&x)[2]) const { x[0] = v.x; x[1] = v.y; }
  __device__ __forceinline__ void get2(f2 (&x)[1]) const { x[0] = f2{v.x, v.y}; }
};

template <int PX>
__device__ __forceinline__ void load_f32(const float* p, float (&x)[PX]) {
  if constexpr (PX % 4 == 0) {
#pragma unroll
    for (int i = 0; i < PX; i += 4) {
      float4 t = *reinterpret_cast<const float4*>(p + i);
      x[i] = t.x; x[i + 1] = t.y; x[i + 2] = t.z; x[i + 3] = t.w;
    }
  } else if constexpr (PX == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    x[0] = t.x; x[1] = t.y;
  } else {
#pragma unroll
    for (int i = 0; i < PX; ++i) x[i] = p[i];
  }
}

// ---- wave / block reductions ----------------------------------------------------------------
// Reductions over the 64 lanes with DPP moves (vector-ALU lane permutes) instead of ds_bpermute: a scan inside every row
// of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of a row into the next row (row_bcast:15, rows 1 and 3) and lane 31 into
// the upper half (row_bcast:31): lane 63 holds the result, which v_readlane hands to every lane.  The LDS crossbar the
// shuffles went through was what a one-workgroup kernel with 16 waves waited for (380 ds_bpermute in the W finish).
// Lanes without a source keep the identity (sum: 0, max: -inf).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v, float ident) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v, double ident) {
  const long long vb = __double_as_longlong(v), ib = __double_as_longlong(ident);
  const int lo = __builtin_amdgcn_update_dpp((int)ib, (int)vb, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ib >> 32), (int)(vb >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double lane63(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v += dpp_move<0x111, 0xf>(v, (T)0);   // row_shr:1
  v += dpp_move<0x112, 0xf>(v, (T)0);   // row_shr:2
  v += dpp_move<0x114, 0xf>(v, (T)0);   // row_shr:4
  v += dpp_move<0x118, 0xf>(v, (T)0);   // row_shr:8
  v += dpp_move<0x142, 0xa>(v, (T)0);   // row_bcast:15 -> rows 1, 3
  v += dpp_move<0x143, 0xc>(v, (T)0);   // row_bcast:31 -> rows 2, 3
  return lane63(v);
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
  const T ninf = -(T)INFINITY;
  T o;
  o = dpp_move<0x111, 0xf>(v, ninf); v = o > v ? o : v;
  o = dpp_move<0x112, 0xf>(v, ninf); v = o > v ? o : v;
  o = dpp_move<0x114, 0xf>(v, ninf); v = o > v ? o : v;
  o = dpp_move<0x118, 0xf>(v, ninf); v = o > v ? o : v;
  o = dpp_move<0x142, 0xa>(v, ninf); v = o > v ? o : v;
  o = dpp_move<0x143, 0xc>(v, ninf); v = o > v ? o : v;
  return lane63(v);
}

// Block-wide reduction of NV doubles (sum for the first NSUM, max for the rest).  `scratch` holds at
// least (blockDim.x / 64 + 1) * NV doubles.  Result valid in thread 0.  Stage 1: shuffles inside each
// wave; stage 2: thread i < NV combines value i over the waves (in wave order - deterministic).
// FMAX: the maxima are fp32 values held in doubles - their wave stage moves one dword per step instead of two.
template <int NV, int NSUM, bool FMAX = false>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (i < NSUM) ? wave_sum(v[i]) : (FMAX ? (double)wave_max((float)v[i]) : wave_max(v[i]));
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    double acc = scratch[i];
    for (int w = 1; w < nw; ++w) {
      const double o = scratch[w * NV + i];
      acc = (i < NSUM) ? acc + o : (o > acc ? o : acc);
    }
    scratch[nw * NV + i] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = scratch[nw * NV + i];
  }
  __syncthreads();
}

// Block-wide reduction of NV fp32 values (sum for the first NSUM, max for the rest) with ONE barrier: fp32 DPP reductions
// inside each wave (6 moves + 6 adds per value, a third of the double version), the per-wave results as doubles in
// `scratch` ((blockDim.x / 64) * NV doubles), then thread i < NV combines value i over the waves in wave order
// (deterministic, fp64) and hands it to emit(i, value) - no gathering thread, no second and third barrier.  The other
// threads leave after the barrier; `scratch` must not be reused before the next barrier of the caller.
// The same in two halves for a caller that has a barrier of its own between them: `block_reduce_f32_wave` (every wave, BEFORE the
// barrier: its sums / maxima into scratch that nothing else uses meanwhile), `block_reduce_f32_finish` (after it: thread i < NV adds
// the waves' values of item i in wave order - the same value as block_reduce_f32 - and emits it; no further barrier).
// The wave stage of NV values at once.  One value at a time (wave_sum / wave_max) is 6 dependent cross-lane steps per value:
// 16 values x 16 waves of the fused kernel's record were 3.5 us of every workgroup's critical path at the headline
// (profiles/r03bc_*: timing without them).  Packed: gfx950's half- and row-exchanges (v_permlane32_swap / v_permlane16_swap) take a PAIR
// of values per instruction - after the swap the lower half (the even rows) holds both halves' copies of the first value, the upper
// half (the odd rows) both copies of the second, one add or max folds them - so two rounds leave a quarter of the values per lane,
// each already combined over the four rows, and only those take the four in-row steps: 8 + 4 exchanges and 4 x 4 row steps for
// 16 values instead of 16 x 6.  The total of slot 4 m + 2 (row & 1) + (row >> 1) ends up in register m of lane 15 of each row.
// Sums and maxima are packed separately (P = 8 or 16 slots each, padded with the operation's identity).
template <int P, bool SUM>
__device__ __forceinline__ void wave_reduce_packed(float (&x)[P]) {   // on return x[m], m < P / 4, lane 15 of row r: total of slot 4 m + 2 (r & 1) + (r >> 1)
  static_assert(P == 8 || P == 16, "8 or 16 slots");
  auto fold = [](float a, float b) { return SUM ? a + b : fmaxf(a, b); };
  float y[P / 2];
#pragma unroll
  for (int j = 0; j < P / 2; ++j) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[2 * j]), __float_as_uint(x[2 * j + 1]), false, false);
    y[j] = fold(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
#pragma unroll
  for (int m = 0; m < P / 4; ++m) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[2 * m]), __float_as_uint(y[2 * m + 1]), false, false);
    float z = fold(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const float id = SUM ? 0.f : -INFINITY;
    z = fold(z, dpp_move<0x111, 0xf>(z, id));   // row_shr:1
    z = fold(z, dpp_move<0x112, 0xf>(z, id));   // row_shr:2
    z = fold(z, dpp_move<0x114, 0xf>(z, id));   // row_shr:4
    z = fold(z, dpp_move<0x118, 0xf>(z, id));   // row_shr:8
    x[m] = z;
  }
}
// (the same contract as block_reduce_f32_wave: the waves' values in scratch[wave * NV + i]; another order of the additions inside a wave)
template <int NV, int NSUM>
__device__ __forceinline__ void block_reduce_f32_wave_packed(const float (&v)[NV], double* scratch) {
  constexpr int NMAX = NV - NSUM;
  constexpr int PS = NSUM <= 8 ? 8 : 16, PM = NMAX <= 8 ? 8 : 16;
  static_assert(NSUM <= 16 && NMAX <= 16, "at most 16 sums and 16 maxima");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s[PS], m[PM];
#pragma unroll
  for (int i = 0; i < PS; ++i) s[i] = i < NSUM ? v[i] : 0.f;
#pragma unroll
  for (int i = 0; i < PM; ++i) m[i] = i < NMAX ? v[NSUM + i] : -INFINITY;
  wave_reduce_packed<PS, true>(s);
  wave_reduce_packed<PM, false>(m);
  if ((lane & 15) == 15) {
    const int row = lane >> 4, c = 2 * (row & 1) + (row >> 1);
#pragma unroll
    for (int q = 0; q < PS / 4; ++q)
      if (4 * q + c < NSUM) scratch[wave * NV + 4 * q + c] = (double)s[q];
#pragma unroll
    for (int q = 0; q < PM / 4; ++q)
      if (4 * q + c < NMAX) scratch[wave * NV + NSUM + 4 * q + c] = (double)m[q];
  }
}
template <int NV, int NSUM>
__device__ __forceinline__ void block_reduce_f32_wave(const float (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float r[NV];
#ifdef ESPM_EXPERIMENT_NO_WAVE_REDUCE   // TIMING ONLY: what the cross-lane stage of the fused kernel's record reduction costs
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = (i < NSUM) ? 64.f * v[i] : v[i];
#else
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = (i < NSUM) ? wave_sum(v[i]) : wave_max(v[i]);
#endif
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = (double)r[i];
  }
}
template <int NV, int NSUM, typename Emit>
__device__ __forceinline__ void block_reduce_f32_finish(const double* scratch, Emit emit) {
  const int nw = blockDim.x >> 6;
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    double o[16];   // (all reads requested together: as a loop over a run-time wave count every read waited for the one before)
#pragma unroll
    for (int w = 0; w < 16; ++w) o[w] = w < nw ? scratch[w * NV + i] : 0.0;
    double acc = o[0];
#pragma unroll
    for (int w = 1; w < 16; ++w)
      if (w < nw) acc = (i < NSUM) ? acc + o[w] : (o[w] > acc ? o[w] : acc);
    emit(i, acc);
  }
}

template <int NV, int NSUM, typename Emit>
__device__ __forceinline__ void block_reduce_f32(const float (&v)[NV], double* scratch, Emit emit) {
  const int nw = blockDim.x >> 6;
  if constexpr (NSUM <= 16 && NV - NSUM <= 16)   // (the wave stage as in the fused kernel: the same order of the additions in both paths)
    block_reduce_f32_wave_packed<NV, NSUM>(v, scratch);
  else
    block_reduce_f32_wave<NV, NSUM>(v, scratch);
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    double acc = scratch[i];
    for (int w = 1; w < nw; ++w) {
      const double o = scratch[w * NV + i];
      acc = (i < NSUM) ? acc + o : (o > acc ? o : acc);
    }
    emit(i, acc);
  }
}

// ---- tail of the local W update: column sums of G W' and rel_W (base.py:323) from the partials and W', W -----------------
// One workgroup of any size (a kernel of its own, or an extra workgroup of the next H-step, which then does not wait for
// it: mu_ell_kernel.hpp).  `scratch`: (blockDim.x / 64 + 1) * (KP + 1) + 1 doubles.  HELD entries of W per thread are
// requested up front, before the partials are reduced.
struct WTailArgs {
  const double* parts;
  const float* w_old;
  const float* w_new;
  double* colsum_gw;
  double* hist_slot;
  double* pg_q;   // projected gradient: sum of the third row of partials goes here, else null
  int n, k, nbk;
  float rel_tol;
};
template <int HELD>
__device__ __forceinline__ void w_tail_body(const WTailArgs& a, double* scratch) {
  const int tid = threadIdx.x, nt = blockDim.x, nwg = a.k * a.nbk, mk = a.n * a.k;
  double* s_mean = scratch + (nt / 64 + 1) * (KP + 1);
  float wn[HELD], wo[HELD];
  if (a.hist_slot) {
#pragma unroll
    for (int u = 0; u < HELD; ++u) {
      const int i = tid + u * nt;
      wn[u] = i < mk ? a.w_new[i] : 1.f;
      wo[u] = i < mk ? a.w_old[i] : 1.f;
    }
  }
  double v[KP + 1];
#pragma unroll
  for (int i = 0; i <= KP; ++i) v[i] = 0.0;
  for (int j = tid; j < a.nbk; j += nt) {
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      if (kk < a.k) {
        v[kk] += a.parts[kk * a.nbk + j];
        v[KP] += a.parts[nwg + kk * a.nbk + j];
      }
  }
  block_reduce<KP + 1, KP + 1>(v, scratch);
  if (tid == 0) {
    for (int kk = 0; kk < KP; ++kk) a.colsum_gw[kk] = kk < a.k ? v[kk] : 0.0;
    *s_mean = v[KP] / (double)mk;
  }
  __syncthreads();
  if (a.pg_q) {
    double q1[1] = {0.0};
    for (int j = tid; j < nwg; j += nt) q1[0] += a.parts[2 * nwg + j];
    block_reduce<1, 1>(q1, scratch);
    if (tid == 0) *a.pg_q = q1[0];
  }
  if (!a.hist_slot) return;
  const float shift = (float)((double)a.rel_tol * *s_mean);
  float rel = 0.f;   // fp32 like the register-resident W finish compares doubles of fp32 values: the quotient of two
                     // fp32 numbers rounded once is within 1 ulp of that; rel_W is a stop-rule statistic (base.py:323)
#pragma unroll
  for (int u = 0; u < HELD; ++u)
    if (tid + u * nt < mk) rel = fmaxf(rel, fabsf(wn[u] - wo[u]) / (wn[u] + shift));
  for (int i0 = tid + HELD * nt; i0 < mk; i0 += 8 * nt) {   // (beyond the held entries: eight loads of each array in flight per trip)
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nt;
      x[u] = i < mk ? a.w_new[i] : 1.f;
      y[u] = i < mk ? a.w_old[i] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) rel = fmaxf(rel, fabsf(x[u] - y[u]) / (x[u] + shift));
  }
  double r1[1] = {(double)rel};
  block_reduce<1, 0>(r1, scratch);
  if (tid == 0) a.hist_slot[ESPM_HI_REL_W] = r1[0];
}

// ---- per-column simplex multiplier ----------------------------------------------------------
// Root of f(nu) = sum_i max(num_i / (nu + den_i), eps) - 1 inside the reference's bracket
// (espm/estimators/dicotomy.py:29-49).
//
// The unknown is re-parametrised as delta = nu + d*, d* = min{den_i : num_i > 0} (the last pole of
// f sits at nu = -d*), and the shifted denominators e_i = den_i - d* >= 0 are formed once.  When the
// simplex forces mass onto a component whose numerator is tiny (e.g. an NNDSVD zero clamped to
// 1e-14) the root lies within ~num_i of that pole: delta ~ 1e-14 is then an ordinary fp32 number,
// whereas nu = -d* + 1e-14 is not representable (the reference resolves it only to ulp(d*) in fp64).
// f is convex and decreasing for delta > 0, so Newton from the left end converges monotonically;
// iterates stay inside the running bracket and fall back to bisection when Newton leaves it or
// stops halving its step.  Outputs delta and e[]: the update is num_i / (delta + e_i).
// Returns false when the reference's preconditions (dicotomy.py:17-19) do not hold.
template <typename T>
__device__ __forceinline__ T fast_rcp(T x);
template <>
__device__ __forceinline__ float fast_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ double fast_rcp<double>(double x) { return 1.0 / x; }

// fast_exit (the H update's per-pixel root, tolerance 1e-6): the evaluation that would only confirm a Newton step is left out when the
// step's PREDICTED residual is inside the tolerance.  A Newton step from x lands at |f| ~ f(x)^2 f''(x) / (2 f'(x)^2) with
// f'' / 2 = sum t inv^2 (accumulated beside f and f'); the exit is taken when four times that prediction is <= tol.  A fixed threshold
// on |f(x)| alone (round 4: 1e-4) is NOT enough: f'' / f'^2 is large where a tiny numerator sits next to its pole, and residuals of
// 7.6e-6 were found there - above the tolerance the confirming evaluation enforces.  The clamp at eps puts kinks into f that
// the prediction does not see, each worth at most eps: the exit is only taken where k eps is far below tol (a log_shift of 1e-14 in
// every fit; the reference's tests with eps = 0.02 take every evaluation).  tests/test_gpu_updates.py::test_per_pixel_root_* holds the
// routine itself (espm_simplex_root_f32) to the tolerance on near-pole inputs with and without the exit.
template <typename T, int K>
__device__ __forceinline__ bool simplex_root(const T (&num)[K], const T (&den)[K], int k, T eps, T tol,
                                             int maxit, T& delta, T (&e)[K], bool fast_exit = false) {
  T nmax = 0, dmin_all = INFINITY, dstar = INFINITY, nsum = 0;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (i < k) {
      ok = ok && (num[i] >= 0) && (den[i] >= 0);
      if (num[i] > 0) dstar = fmin(dstar, den[i]);
      nmax = fmax(nmax, num[i]);
      dmin_all = fmin(dmin_all, den[i]);
      nsum += num[i];
    }
  }
  ok = ok && (nsum > 0) && (nsum < (T)INFINITY);
  if (!ok) {
    delta = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) e[i] = i < k ? den[i] : (T)1;
    return false;
  }
  T lo = 0;  // = max_i (num_i / 2 - den_i) + d*   (dicotomy.py:29-43)
#pragma unroll
  for (int i = 0; i < K; ++i) {
    e[i] = i < k ? den[i] - dstar : (T)1;
    if (i < k && num[i] > 0) lo = fmax(lo, num[i] / 2 - e[i]);
  }
  T hi = (T)(2 * k) * nmax + (dstar - dmin_all);  // dicotomy.py:49 (in this order: numerators of 1e-13 must survive next to den ~ 30)
  // start at nu = 0 (delta = d*) when that lies inside the bracket: multiplicative updates sit close to their
  // fixed point, where sum_i num_i / den_i is already ~1; the safeguards below handle either side of the root
  T x = fmax(lo, fmin(dstar, hi)), dxold = hi - lo;
  const bool fast = fast_exit && (T)(4 * K) * eps <= (T)1e-3 * tol;
  for (int it = 0; it < maxit; ++it) {
    T f = -1, fp = 0, fpp = 0;   // f, f', f'' / 2
#pragma unroll
    for (int i = 0; i < K; ++i) {
      if (i < k) {
        T inv = fast_rcp<T>(x + e[i]);
        T t = num[i] > 0 ? num[i] * inv : (T)0;
        if (t > eps) {
          const T u = t * inv;
          f += t;
          fp -= u;
          if (fast_exit) fpp += u * inv;
        } else {
          f += eps;
        }
      }
    }
    if (fabs(f) <= tol) break;
    if (f > 0) lo = x; else hi = x;
    // Newton step, replaced by bisection when it leaves the bracket or stops halving the step
    T dx = fp < 0 ? -f * fast_rcp<T>(fp) : (T)0;   // (a Newton step tolerates a 1 ulp reciprocal)
    T xn = x + dx;
    bool newton = true;
    if (!(fp < 0) || !(xn > lo && xn < hi) || fabs(dx) > (T)0.5 * fabs(dxold)) {
      dx = (hi - lo) / 2;
      xn = lo + dx;
      newton = false;
    }
    dxold = dx;
    if (xn == x) break;
    x = xn;
    if (fast && newton && (T)4 * f * f * fpp <= tol * fp * fp) break;
  }
  delta = x;
  return true;
}

// ---- quadratic surrogate of the Laplacian term (algo = "l2_surrogate") ----------------------------------------
// H' is the positive root of a H'^2 + (b + nu) H' - c = 0 (updates.py:285-300): g(s, q) = sqrt(s^2 + q) - s with
// s = b + nu and q = 4 a c, formed without cancellation, gives H' = g / (2 a).
// (v_sqrt_f32 / v_rcp_f32: 1 ulp each, and far fewer registers than the correctly rounded expansions)
__device__ __forceinline__ float hq_root(float s, float q) {
  const float r = __builtin_amdgcn_sqrtf(fmaf(s, s, q));
  return s >= 0.f ? q * __builtin_amdgcn_rcpf(r + s) : r - s;
}
// nu with sum_k max(g(b_k + nu, 4 a c_k), 2 a eps) = 2 a (dicotomy.py:57-82: same bracket), by a Newton iteration kept
// inside the running bracket; the sum decreases with nu.  Returns false when the preconditions (a > 0, c >= 0) fail.
template <int K>
__device__ __forceinline__ bool simplex_root_hq(float a, const float (&b)[K], const float (&c)[K], float eps, int maxit, float& nu) {
  bool ok = a > 0.f;
  float bmax = -INFINITY, bsum = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    ok = ok && c[i] >= 0.f && b[i] == b[i];
    bmax = fmaxf(bmax, b[i] * b[i] * __builtin_amdgcn_rcpf(a) + 2.f * a + 2.f * (b[i] + c[i]));
    bsum += b[i];
  }
  nu = 0.f;
  if (!ok) return false;
  float hi = (float)K * bmax * 1.5f + 1e-3f;              // sum < 2 a there
  float lo = -(2.f * a + bsum) * (1.1f / (float)K) - 1e-3f;  // sum > 2 a there
  const float target = 2.f * a, floor_g = 2.f * a * eps;
  float x = fminf(fmaxf(0.f, lo), hi), dxold = hi - lo;
  for (int it = 0; it < maxit; ++it) {
    float f = -target, fp = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float s = b[i] + x, q = 4.f * a * c[i];
      const float r = __builtin_amdgcn_sqrtf(fmaf(s, s, q));
      const float g = s >= 0.f ? q * __builtin_amdgcn_rcpf(r + s) : r - s;
      if (g > floor_g) {
        f += g;
        fp -= g * __builtin_amdgcn_rcpf(r);   // d g / d nu = s / r - 1 = -g / r
      } else {
        f += floor_g;
      }
    }
    if (fabsf(f) <= 4e-7f * target) break;
    if (f > 0.f) lo = x; else hi = x;
    float dx = fp < 0.f ? -f * __builtin_amdgcn_rcpf(fp) : 0.f;
    float xn = x + dx;
    if (!(fp < 0.f) || !(xn > lo && xn < hi) || fabsf(dx) > 0.5f * fabsf(dxold)) {
      dx = (hi - lo) * 0.5f;
      xn = lo + dx;
    }
    dxold = dx;
    if (xn == x) break;
    x = xn;
  }
  nu = x;
  return true;
}

// ---- 5-point graph Laplacian (espm/utils.py:39-76) as a stencil on the local row block ------
// (H L)[q] = deg(q) H[q] - sum of the existing 4-neighbours; rows above/below the local block
// come from halo_top / halo_bot when present (sharded image), otherwise the block edge is the
// image edge (zero-flux boundary).
__device__ __forceinline__ float stencil_hl(const float* hrow, const float* halo_top, const float* halo_bot,
                                            int q, int nx, int ny, float hc) {
  const int i = q / ny, j = q - i * ny;
  float acc = 0.f;
  int deg = 0;
  if (j > 0) { acc += hrow[q - 1]; ++deg; }
  if (j < ny - 1) { acc += hrow[q + 1]; ++deg; }
  if (i > 0) { acc += hrow[q - ny]; ++deg; }
  else if (halo_top) { acc += halo_top[j]; ++deg; }
  if (i < nx - 1) { acc += hrow[q + ny]; ++deg; }
  else if (halo_bot) { acc += halo_bot[j]; ++deg; }
  return (float)deg * hc - acc;
}

// kernel-side argument blocks and launchers (mu_h_step.hip, mu_w_step.hip, mu_aux.hip)
struct HStepArgs {
  const void* x_cm;
  const void* x_pm;  // pixel-major copy (p, n_pad): the matrix-core H-step of the wide build streams this one
  int mfma;          // matrix-core kernels allowed (espm_mu_state.no_fused == 0)
  const float* gw_s;
  const double* colsum_gw;
  const float* h_in;
  float* h_out;
  float* h_t;
  const float* mu;
  const float* fixed_h;
  const float* halo_top;
  const float* halo_bot;
  const double* hstat_in;
  double* hpart;
  int n, k, p, nx, ny, p_pad;
  int x_tile;        // pixel-block width of the tile-major x_cm
  int n_cm;          // channel rows per pixel block of x_cm
  const void* gw_a;  // reserved (null)
  const float* gw_p; // reserved (null)
  int simplex_h, grid_mode, compute_loss, write_h;
  int have_prev;     // h_out still holds the H that preceded h_in: evaluate rel_H (base.py:324)
  float lambda_l, sigma_l, eps_reg, log_shift, tol, xscale, rel_tol;
  double inv_count;  // 1 / (k * p_total)
  // sparse count store (mu_ell_kernel.hpp)
  const uint32_t* ell;
  const int32_t* ell_off;
  const float* ell_klc;
  const int32_t* ell_pix;
  int ell_bits, n_pad;
  int ell_tp;        // pixels per workgroup of the sparse H-step (= tile_px: 64, 128, 256 or 512)
  const float* l2_m; // Frobenius branch: (KP, KP) GW^T GW, else null
  const float* breg_sr; // Bregman variant: per-pixel sums of the stored X (p_pad), else null
  int h_rule;        // 0: log surrogate (multiplicative_step_h), 1: quadratic surrogate (multiplicative_step_hq)
  const float* fill_num;  // sparse store: numerators of the pixels that hold nothing but the fill (k, fill_n), else null
  int fill_n;
  // espm_mu_iterate, sparse store, local W update: the tail of the previous W update rides in this launch as one extra
  // workgroup (tail_on = 1; 2: no extra workgroup - the launch's own workgroups share the tail, mu_fused_kernel.hpp), and the workgroups sum the partial column sums of G W' themselves (cs_parts: (k, cs_nbk) doubles,
  // into k doubles of LDS at byte offset cs_lds_off) instead of waiting for colsum_gw
  const double* cs_parts;
  int cs_nbk, cs_lds_off, tail_on;
  WTailArgs tail;
  // fused half-steps (mu_fused_kernel.hpp): a workgroup covers TWO record slots of hpart - its record goes to slot
  // 2 * blockIdx.x of rec_nb, zeros (neutral for every field) to the next; 0: one record per workgroup
  int rec_nb;
};
struct HFinalizeArgs {
  const double* hpart;
  const double* colsum_gw;
  const double* hstat_in;
  double* hstat_out;
  double* hist_slot;
  int nblk, k, compute_loss, have_prev;
  float xscale;
  double* pg_q;  // projected-gradient rule: where the sum of the records' PGQ field goes, else null
};
// index bits of a W-step entry of the sparse store: log2 of the pixels per block (64 .. 1024)
__host__ __device__ inline int ell_pbits(int pb) {
  int b = 6;
  while ((1 << b) < pb) ++b;
  return b;
}
struct WAccumArgs {
  const void* x_pm;
  const void* x_cm;  // tile-major X (the matrix-core kernel of the wide build streams this copy)
  int x_tile, n_cm, p_pad, mfma;
  const float* gw_s;
  const float* h_t;
  float* a_slab;
  int n_pad, p, ppb;
  // sparse count store (mu_ell_kernel.hpp)
  const uint32_t* ell;
  const int32_t* ell_off;
  const int32_t* chan_perm;
  int n_cg;
  int pb, pbits;     // pixels per block (espm_mu_state.ell_pb) and its log2: index bits of an entry
  int l2;            // Frobenius branch: A = X H^T
};
struct WFinishArgs {
  const float* g;
  const float* g_t;     // (m, n_pad) transposed copy of g or null
  const float* colsum_g;
  const float* w_old;
  float* w_new;
  const float* a;       // (k, n_pad): R H^T summed over the pixel blocks (and over ranks when sharded)
  const double* hstat;
  const float* fixed_w;
  const int32_t* simplex_rows;
  float* scratch;
  const float* breg_sr; // Bregman variant: per-channel sums of the stored X (n), else null
  float pg_gamma_w;     // > 0: projected-gradient step W - grad / gamma (updates.py:353-370)
  double* pg_q;         // its linesearch term sum <W' - W, grad> + gamma ||W' - W||^2, else null
  float* gw_s;
  double* colsum_gw;
  void* gw_a;    // reserved (null)
  float* gw_p;   // reserved (null)
  double* hist_slot;
  int n, m, k, n_pad, n_cm, simplex_w, update_w;
  float log_shift, tol, rel_tol, xscale, gw_floor;
};


// ---- reduction of the H-step's per-workgroup records (one workgroup of 256 threads) ---------------
// (The one-workgroup form: the definition of the order of operations.  What the launches run since the end of round 3 is
//  h_finalize_one below - one workgroup per value, same order, same bits.)
__device__ __forceinline__ void h_finalize_body(const HFinalizeArgs& a, double* scratch) {
  // 256 threads; records are field-major (hpart[field][block]) so every load is coalesced, and U blocks
  // per thread are in flight at once.  Few waves on purpose: the cross-lane part costs per wave.
  constexpr int NV = ESPM_HP_NSCALAR + 2 * KP + 1;   // [0..3] scalar sums, [4..4+KP) row sums | [4+KP] RELH, then maxima, RELW
  constexpr int V_RELH = 4 + KP, V_MAX = 5 + KP, V_RELW = 5 + 2 * KP;
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  v[V_RELW] = -1.0;   // (-1: the launch carried no W update's tail; a record's share of rel_W is >= 0)
  const size_t nb = a.nblk;
  // (the wide build's records have 37 fields: 4 of them in flight are 296 registers on top of the 74 of the sums, which took
  //  every kernel this body rides in - the slab reductions - to one wave per SIMD; the order of the sums does not depend on U)
  constexpr int U = KP > 8 ? 1 : 4;
  for (int b0 = threadIdx.x; b0 < a.nblk; b0 += U * 256) {
    double t[U][ESPM_HP_RELH + 1], tw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u * 256;
#pragma unroll
      for (int i = 0; i <= ESPM_HP_RELH; ++i) t[u][i] = b < a.nblk ? a.hpart[i * nb + b] : 0.0;
      tw[u] = b < a.nblk ? a.hpart[(size_t)ESPM_HP_RELW * nb + b] : -1.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < 4 + KP; ++i) v[i] += t[u][i];
      v[V_RELH] = fmax(v[V_RELH], t[u][ESPM_HP_RELH]);
      v[V_RELW] = fmax(v[V_RELW], tw[u]);
#pragma unroll
      for (int i = 0; i < KP; ++i) v[V_MAX + i] = fmax(v[V_MAX + i], t[u][ESPM_HP_MAX + i]);
    }
  }
  block_reduce<NV, 4 + KP>(v, scratch);
  if (a.pg_q) {  // (uniform) the quadratic bound of the projected gradient's linesearch
    double q1[1] = {0.0};
    for (int b = threadIdx.x; b < a.nblk; b += 256) q1[0] += a.hpart[(size_t)ESPM_HP_PGQ * nb + b];
    block_reduce<1, 1>(q1, scratch);
    if (threadIdx.x == 0) *a.pg_q = q1[0];
  }
  if (threadIdx.x == 0) {
    double sumy = 0.0;
    for (int kk = 0; kk < a.k; ++kk) sumy += a.colsum_gw[kk] * a.hstat_in[ESPM_HS_ROWSUM + kk];
    if (a.compute_loss) a.hist_slot[ESPM_HI_KLX] = (double)a.xscale * 0.6931471805599453 * v[ESPM_HP_KL];
    a.hist_slot[ESPM_HI_REG] = v[ESPM_HP_REG];
    a.hist_slot[ESPM_HI_LAP] = v[ESPM_HP_LAP];
    a.hist_slot[ESPM_HI_SUMY] = sumy;
    a.hist_slot[ESPM_HI_BAD] = v[ESPM_HP_BAD];
    if (a.have_prev) a.hist_slot[ESPM_HI_REL_H] = v[V_RELH];
    if (v[V_RELW] >= 0.0) a.hist_slot[ESPM_HI_REL_W] = v[V_RELW];   // rel_W of the update that produced this state, where the H-step's workgroups formed it (mu_fused_kernel.hpp)
    if (a.hstat_out) {
      for (int kk = 0; kk < KP; ++kk) {
        a.hstat_out[ESPM_HS_ROWSUM + kk] = v[ESPM_HP_ROWSUM + kk];
        a.hstat_out[ESPM_HS_MAX + kk] = v[V_MAX + kk];
      }
    }
  }
}

// ONE value of h_finalize_body by a whole workgroup of 256 threads, and the outputs that depend on that value alone.  Nothing in
// h_finalize_body combines two reduced values (every entry of the history slot and of hstat is one of them, SUMY needs none), so
// the record reduction splits into H_FINALIZE_JOBS independent jobs: job i < NV reduces v[i] of the body - the same order of
// operations (a thread's blocks 256 apart in ascending order, the wave stage, the waves in order), hence the same bits -, job NV
// forms SUMY, job NV + 1 the projected gradient's quadratic bound.  As ONE workgroup the body is a launch's tail: at the headline it
// ends 2.6 us after the workgroups that reduce the slabs, on a 64-row shard 3.7 us (profiles/r03ay_*); as 23 workgroups of two
// loads and one reduction each it ends before them.
constexpr int H_FINALIZE_NV = ESPM_HP_NSCALAR + 2 * KP + 1;
constexpr int H_FINALIZE_JOBS = H_FINALIZE_NV + 2;
__device__ __forceinline__ void h_finalize_one(const HFinalizeArgs& a, int job, double* scratch) {
  constexpr int NV = H_FINALIZE_NV, V_RELH = 4 + KP, V_MAX = 5 + KP, V_RELW = 5 + 2 * KP;
  const size_t nb = a.nblk;
  if (job == NV) {   // sum_k colsum(GW)_k rowsum(H)_k of the INPUT state (hstat_in): no records
    if (threadIdx.x == 0) {
      double sumy = 0.0;
      for (int kk = 0; kk < a.k; ++kk) sumy += a.colsum_gw[kk] * a.hstat_in[ESPM_HS_ROWSUM + kk];
      a.hist_slot[ESPM_HI_SUMY] = sumy;
    }
    return;
  }
  if (job == NV + 1) {
    if (a.pg_q) {  // (uniform) the quadratic bound of the projected gradient's linesearch
      double q1[1] = {0.0};
      for (int b = threadIdx.x; b < a.nblk; b += 256) q1[0] += a.hpart[(size_t)ESPM_HP_PGQ * nb + b];
      block_reduce<1, 1>(q1, scratch);
      if (threadIdx.x == 0) *a.pg_q = q1[0];
    }
    return;
  }
  const bool is_sum = job < 4 + KP;
  const int field = is_sum ? job : (job == V_RELH ? ESPM_HP_RELH : (job == V_RELW ? ESPM_HP_RELW : ESPM_HP_MAX + (job - V_MAX)));
  double v[1] = {job == V_RELW ? -1.0 : 0.0};
  for (int b = threadIdx.x; b < a.nblk; b += 256) {
    const double t = a.hpart[(size_t)field * nb + b];
    v[0] = is_sum ? v[0] + t : fmax(v[0], t);
  }
  if (is_sum)
    block_reduce<1, 1>(v, scratch);
  else
    block_reduce<1, 0>(v, scratch);
  if (threadIdx.x != 0) return;
  if (job == ESPM_HP_KL) {
    if (a.compute_loss) a.hist_slot[ESPM_HI_KLX] = (double)a.xscale * 0.6931471805599453 * v[0];
  } else if (job == ESPM_HP_REG) {
    a.hist_slot[ESPM_HI_REG] = v[0];
  } else if (job == ESPM_HP_LAP) {
    a.hist_slot[ESPM_HI_LAP] = v[0];
  } else if (job == ESPM_HP_BAD) {
    a.hist_slot[ESPM_HI_BAD] = v[0];
  } else if (is_sum) {
    if (a.hstat_out) a.hstat_out[ESPM_HS_ROWSUM + (job - ESPM_HP_ROWSUM)] = v[0];
  } else if (job == V_RELH) {
    if (a.have_prev) a.hist_slot[ESPM_HI_REL_H] = v[0];
  } else if (job == V_RELW) {
    if (v[0] >= 0.0) a.hist_slot[ESPM_HI_REL_W] = v[0];
  } else {
    if (a.hstat_out) a.hstat_out[ESPM_HS_MAX + (job - V_MAX)] = v[0];
  }
}


// argument blocks from the public state (shared by the C ABI and the tuning harness)
inline HStepArgs make_h_args(const espm_mu_state* st, int src, int write_h) {
  HStepArgs a;
  a.x_cm = st->x_cm;
  a.x_pm = st->x_pm;
  a.mfma = st->no_fused == 0;   // (no_fused != 0: the vector-ALU kernels, for A/B)
  a.gw_s = st->gw_s;
  a.colsum_gw = st->colsum_gw;
  a.h_in = st->h[src];
  a.h_out = st->h[1 - src];
  a.h_t = st->h_t;
  a.mu = st->mu;
  a.fixed_h = st->fixed_h;
  a.halo_top = st->grid_mode ? st->halo_top : nullptr;
  a.halo_bot = st->grid_mode ? st->halo_bot : nullptr;
  a.hstat_in = st->hstat[src];
  a.hpart = st->hpart;
  a.n = st->n;
  a.k = st->k;
  a.p = st->p;
  a.nx = st->nx;
  a.ny = st->ny;
  a.p_pad = st->p_pad;
  a.x_tile = st->x_tile;
  a.n_cm = st->n_cm;
  a.gw_a = st->gw_a;
  a.gw_p = st->gw_p;
  a.simplex_h = st->simplex_h;
  a.grid_mode = st->grid_mode;
  a.compute_loss = st->compute_loss;
  a.write_h = write_h;
  a.have_prev = st->it > 0;  // h[1-src] then still holds the H that preceded h[src]
  a.rel_tol = st->rel_tol;
  a.inv_count = 1.0 / ((double)st->k * (double)(st->p_total > 0 ? st->p_total : st->p));
  a.lambda_l = st->lambda_l;
  a.sigma_l = st->sigma_l;
  a.eps_reg = st->eps_reg;
  a.log_shift = st->log_shift;
  a.tol = st->dicotomy_tol;
  a.xscale = st->xscale;
  a.ell = st->ell_h;
  a.ell_off = st->ell_h_off;
  a.ell_klc = st->ell_klc;
  a.ell_pix = st->pix_perm;
  a.ell_bits = st->ell_cbits;
  a.ell_tp = st->tile_px;
  a.l2_m = nullptr;
  a.breg_sr = st->breg_sr_px;
  a.h_rule = st->h_rule;
  a.fill_num = (st->x_dtype == ESPM_X_ELL && st->ell_fill_n > 0) ? st->ell_fill_num : nullptr;
  a.fill_n = st->ell_fill_n;
  a.cs_parts = nullptr;
  a.cs_nbk = a.cs_lds_off = a.tail_on = 0;
  a.rec_nb = 0;
  a.n_pad = st->n_pad;
  return a;
}

inline WAccumArgs make_w_args(const espm_mu_state* st) {
  WAccumArgs a;
  a.x_pm = st->x_pm;
  a.x_cm = st->x_cm;
  a.x_tile = st->x_tile;
  a.n_cm = st->n_cm;
  a.p_pad = st->p_pad;
  a.mfma = st->no_fused == 0;   // (no_fused != 0: the vector-ALU kernels, for A/B)
  a.gw_s = st->gw_s;
  a.h_t = st->h_t;
  a.a_slab = st->a_slab;
  a.n_pad = st->n_pad;
  a.p = st->p;
  a.ppb = (st->p + st->nblk_w - 1) / st->nblk_w;
  a.ell = st->ell_w;
  a.ell_off = st->ell_w_off;
  a.chan_perm = st->chan_perm;
  a.n_cg = st->n_cg;
  a.pb = st->ell_pb > 0 ? st->ell_pb : ESPM_ELL_PB;
  a.pbits = ell_pbits(a.pb);
  a.l2 = 0;
  return a;
}

int dispatch_h_step(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream);
int launch_h_finalize(const HFinalizeArgs& args, hipStream_t stream);
int launch_h_ell(const HStepArgs& args, int nblk, hipStream_t stream);
int launch_fused_ell(const HStepArgs& h, const WAccumArgs& w, int nblk, hipStream_t stream, int static_units = 0, int stream_lists = 0);
size_t fused_ell_lds_bytes(int n_pad, int k, int pb);
int launch_ell_count(const uint8_t* x_pm, int n, int n_pad, int p, int p_pad, int cbits, int n_cg, int nblk, int pb,
                     int32_t* cnt_px, int32_t* cnt_bc, float* klc, hipStream_t stream, uint8_t* bkt_px = nullptr, uint8_t* bkt_bc = nullptr);
int launch_ell_plan(const int32_t* cnt_px, const int32_t* cnt_bc, int n, int n_cg, int nblk, int p_pad, int win,
                    int32_t* chan_perm, int32_t* pix_perm, int32_t* h_off, int32_t* w_off, long long* rows, hipStream_t stream);
int launch_ell_fill(const uint8_t* x_pm, int n, int n_pad, int p, int p_pad, int cbits, int n_cg, int nblk, int win, int pb,
                    const int32_t* chan_perm, const int32_t* pix_perm, const int32_t* h_off, const int32_t* w_off,
                    uint32_t* ell_h, uint32_t* ell_w, hipStream_t stream, const uint8_t* x_cm = nullptr, int n_cm = 0, const uint8_t* bkt_px = nullptr,
                    const uint8_t* bkt_bc = nullptr);
int launch_w_ell(const WAccumArgs& args, int k, int nblk, hipStream_t stream);
int launch_ell_fill_num(const float* gw_s, const float* h_in, const int32_t* fill_px, int fill_n, int n, int k, int p_pad, float fill,
                        float* fill_num, hipStream_t stream);
int dispatch_w_accum(const WAccumArgs& args, int k, int x_dtype, int nblk, hipStream_t stream);
int launch_w_reduce(const float* slab, float* out, int nblk, int total, const HFinalizeArgs* fused_finalize,
                    hipStream_t stream, const float* bw_old = nullptr, double* bparts = nullptr, int n = 0, int k = 0, int n_pad = 0);
int launch_w_simplex_update(const WFinishArgs& f, float* a_inout, const double* bparts, double tol, hipStream_t stream, WTailArgs* defer_tail);
int launch_w_reduce_pack(const float* slab, int nblk, int k, int n_pad, const HFinalizeArgs& fin_to_record,
                         const float* h_new, int nx, int ny, int p_pad, int with_halo, void* rec, hipStream_t stream);
int launch_w_finish(const WFinishArgs& args, hipStream_t stream);
int launch_gram(const float* m, int rows, int k, double* part, int part_cap, float* out, hipStream_t stream);
int launch_w_finish_l2(const float* a, int n, int n_pad, int m, int k, const float* g, const float* gtg, const float* hh, const float* w_old,
                       float* w_new, const float* fixed_w, float log_shift, hipStream_t stream);
int launch_w_reduce_update(const WFinishArgs& f, const void* src, size_t src_stride, int nsrc, float* a_out,
                           const double* hpart, int nblk_h, const double* hstat_rs, size_t rec_hstat_off, double* hstat_out,
                           const HFinalizeArgs* fused_finalize, hipStream_t stream, WTailArgs* defer_tail = nullptr);
int launch_w_update_tail(const WTailArgs& t, hipStream_t stream);
int launch_w_exchange_update(const WFinishArgs& f, const void* slabs, size_t slab_stride, int nslab, float* a_out, double* hstat_out,
                             const HFinalizeArgs& fin, const struct ::espm_xchg* xc, unsigned int seq, const float* h_new, int nx, int ny,
                             int p_pad, int with_halo, hipStream_t stream, WTailArgs* defer_tail, double* simplex_bparts = nullptr);
bool w_gsplit_applies(const WFinishArgs& args);   // the W finish with a dictionary G as many-workgroup launches (mu_w_step.hip)
int launch_w_gxchg_update(const WFinishArgs& f, const struct ::espm_xchg* xc, unsigned int seq, const double* hstat_local, double* hstat_out,
                          const float* h_new, int nx, int ny, int p_pad, int with_halo, hipStream_t stream);
WTailArgs make_w_tail_args(const WFinishArgs& f);
int launch_pack_x(const void* src, int src_dtype, int src_layout, int64_t ld, int n, int p, void* x_cm, void* x_pm,
                  int x_dtype, int n_pad, int p_pad, int x_tile, int n_cm, hipStream_t stream);
int launch_hstat(const float* h, int k, int p, int p_pad, double* out, hipStream_t stream);
int launch_dichotomy(const double* num, const double* den, int k, int p, int den_cols, double eps, double tol,
                     int maxit, double* nu_out, int32_t* status, hipStream_t stream);
int launch_shard_pack(const float* a, const double* hstat, const float* h_new, int k, int n_pad, int nx, int ny,
                      int p_pad, int with_halo, void* rec, hipStream_t stream);
int launch_shard_combine(const void* recs, int world, size_t stride, int na, float* a_out, double* hstat_out,
                         hipStream_t stream, const float* bw_old = nullptr, double* bparts = nullptr, int n = 0, int k = 0, int n_pad = 0);
int launch_simplex_root_f32(const float* num, const float* den, int k, int p, float eps, float tol, int maxit, int fast_exit, float* delta_out,
                            float* e_out, int32_t* status, hipStream_t stream);
int launch_dichotomy_acc(double a, const double* b, const double* c, int k, int p, int b_cols, double eps, double tol, int maxit,
                         double* nu_out, int32_t* status, hipStream_t stream);
int launch_dichotomy_pg(const double* a, int k, int p, double eps, double tol, int maxit, double* nu_out, hipStream_t stream);
int launch_laplacian(const float* h, int k, int nx, int ny, int64_t ld, float* out, hipStream_t stream);
int launch_linesearch_terms(const float* h_old, const float* h_new, int k, int p, int p_pad, int nx, int ny, int grid_mode,
                            const float* old_top, const float* old_bot, const float* new_top, const float* new_bot,
                            double* part, double* out, hipStream_t stream);

}  // namespace espm
