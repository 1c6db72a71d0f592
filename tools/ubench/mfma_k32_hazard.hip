// Stand-alone reproducer for the run-to-run differences of gfx950's 32-slot bf16 matrix instruction (DESIGN.md
// section 4, mu_w_mfma_kernel.hpp).  The dependency chain of the wide build's H-step, with nothing around it:
//
//     y   = MFMA(GW split hi|lo, H split hi|lo)            fresh accumulator, 16 x 16 tile, fp32
//     r   = x * rcp(max(y, tiny))                          first reader a compiler-known VALU instruction, the reciprocal in-place asm
//     acc = MFMA(R split hi|lo, GW^T split hi|lo, acc)     running accumulator
//
// in the two forms the library can build it in - FORM 16: three v_mfma_f32_16x16x16_bf16 per product (ah bl, al bh, ah bh), FORM 32:
// two v_mfma_f32_16x16x32_bf16 ([ah | al] x [bl | bl], [ah | al] x [bh | bh]) - launched at ONE and at TWO waves per SIMD (the second
// by letting two 256-thread workgroups share a CU: the same code, an LDS request that does or does not leave room for a second
// workgroup), REPS times from the same inputs, every output compared with the first launch's bit for bit.  Knobs (compile time):
//   -DNOPS=n     n extra wait states (s_nop) between a matrix instruction and the first read of its result, on top of the compiler's
//   -DHOLD=1     the operands of a matrix instruction kept allocated (not overwritten) until its result has been read
//   -DCHAINS=c   independent chains interleaved per wave (the kernel has 4 sub-tiles in flight)
//   -DWITH_LOG=0 without the loss term's in-place v_log_f32 (second transcendental of the chain)
// and hipcc's `-mllvm -amdgpu-mfma-padding-ratio=100` (s_nops in front of every dependent matrix instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_k32_hazard tools/ubench/mfma_k32_hazard.hip ; run: ./tools/ubench/mfma_k32_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#ifndef NOPS
#define NOPS 0
#endif
#ifndef HOLD
#define HOLD 0
#endif
#ifndef CHAINS
#define CHAINS 4
#endif
#ifndef WITH_LOG   // the loss term of the H-step: an in-place v_log_f32 of the ratio behind the reciprocal, summed per lane
#define WITH_LOG 1
#endif

typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                  \
  do {                                                                            \
    hipError_t err_ = (e);                                                        \
    if (err_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

// fp32 -> bf16 hi (truncation) + bf16 lo (the remainder): x = hi + lo to 16 bits, as the library splits its operands
__device__ __forceinline__ void split(const float (&x)[4], s4& hi, s4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u = __float_as_uint(x[i]);
    const float h = __uint_as_float(u & 0xffff0000u);
    hi[i] = (short)(u >> 16);
    lo[i] = (short)(__float_as_uint(x[i] - h) >> 16);
  }
}

template <int FORM>
__device__ __forceinline__ f4 mma3(const s4 ah, const s4 al, const s4 bh, const s4 bl, f4 c) {
  if constexpr (FORM == 32) {
    const b8 a = __builtin_bit_cast(b8, __builtin_shufflevector(ah, al, 0, 1, 2, 3, 4, 5, 6, 7));
    const b8 b1 = __builtin_bit_cast(b8, __builtin_shufflevector(bl, bl, 0, 1, 2, 3, 4, 5, 6, 7));
    const b8 b2 = __builtin_bit_cast(b8, __builtin_shufflevector(bh, bh, 0, 1, 2, 3, 4, 5, 6, 7));
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c, 0, 0, 0);
#if HOLD
    asm volatile("" : : "v"(a), "v"(b1), "v"(b2), "v"(c));
#endif
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
#if HOLD
    asm volatile("" : : "v"(ah), "v"(al), "v"(bh), "v"(bl), "v"(c));
#endif
  }
#if NOPS > 0
  // extra wait states before anything reads c (the asm names c, so it cannot be moved in front of the matrix instructions)
  asm volatile("s_nop %1" : "+v"(c) : "n"(NOPS - 1));
#endif
  return c;
}

// One wave = one 16 x 16 tile chain, CHAINS of them interleaved; `steps` rounds.  g / h / x: per-lane inputs (read once, kept in registers).
template <int FORM>
__global__ __launch_bounds__(256, 2) void chain_kernel(const float* __restrict__ g, const float* __restrict__ h, const float* __restrict__ x,
                                                      float* __restrict__ out, int steps) {
  extern __shared__ float pad_lds[];   // only its SIZE matters: it decides how many workgroups share a CU
  const int t = blockIdx.x * 256 + threadIdx.x;
  s4 gh[CHAINS], gl[CHAINS], g2h[CHAINS], g2l[CHAINS], hh, hl;
  float xv[CHAINS][4];
  {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = h[(size_t)t * 4 + i];
    split(v, hh, hl);
  }
#pragma unroll
  for (int j = 0; j < CHAINS; ++j) {
    float v[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = g[((size_t)t * CHAINS + j) * 4 + i];
      w[i] = g[((size_t)t * CHAINS + (CHAINS - 1 - j)) * 4 + (3 - i)];
      xv[j][i] = x[((size_t)t * CHAINS + j) * 4 + i];
    }
    split(v, gh[j], gl[j]);
    split(w, g2h[j], g2l[j]);
  }
  f4 acc[CHAINS];
  float kl = 0.f;
#pragma unroll
  for (int j = 0; j < CHAINS; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) {
      f4 y = mma3<FORM>(gh[j], gl[j], hh, hl, f4{0.f, 0.f, 0.f, 0.f});
      float r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float yi = fmaxf(y[i], 1e-37f);                                  // a compiler-known first reader (it owes the wait states)
        asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0" : "+v"(yi));          // in place, like the library's
        r[i] = fmaf(xv[j][i], yi, 1e-37f);
#if WITH_LOG
        float lg = r[i];
        asm volatile("v_log_f32 %0, %0\n\ts_nop 0" : "+v"(lg));
        kl = fmaf(xv[j][i], lg, kl);
#endif
      }
      s4 rh, rl;
      split(r, rh, rl);
      acc[j] = mma3<FORM>(rh, rl, g2h[j], g2l[j], acc[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < CHAINS; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[((size_t)t * CHAINS + j) * 4 + i] = acc[j][i] + (i == 0 && j == 0 ? kl : 0.f);
}

template <int FORM>
static void run(const char* name, size_t lds_bytes, int blocks, int steps, int reps, const float* g, const float* h, const float* x, float* out,
                std::vector<float>& first, std::vector<float>& cur) {
  const size_t n = (size_t)blocks * 256 * CHAINS * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_kernel<FORM>, 256, lds_bytes));
  long differing_launches = 0, differing_values = 0;
  float worst = 0.f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipMemset(out, 0xff, n * sizeof(float)));
    hipLaunchKernelGGL(chain_kernel<FORM>, dim3(blocks), dim3(256), lds_bytes, 0, g, h, x, out, steps);
    CHECK(hipGetLastError());
    CHECK(hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost));
    if (r == 0) {
      first = cur;
      continue;
    }
    long d = 0;
    for (size_t i = 0; i < n; ++i)
      if (memcmp(&cur[i], &first[i], 4) != 0) {
        ++d;
        const float rel = fabsf(cur[i] - first[i]) / fmaxf(fabsf(first[i]), 1e-30f);
        if (rel > worst) worst = rel;
      }
    differing_values += d;
    differing_launches += d != 0;
  }
  printf("%-34s workgroups per CU (occupancy query) %d: %ld of %d launches differ from the first, %ld values in all, worst relative %.2e\n", name,
         per_cu, differing_launches, reps - 1, differing_values, worst);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40, steps = argc > 2 ? atoi(argv[2]) : 400;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = 4 * prop.multiProcessorCount;   // enough workgroups to fill every CU twice over
  const size_t nt = (size_t)blocks * 256;
  std::vector<float> hg(nt * CHAINS * 4), hh(nt * 4), hx(nt * CHAINS * 4);
  uint32_t seed = 12345u;
  auto rnd = [&]() {
    seed = seed * 1664525u + 1013904223u;
    return (float)((seed >> 8) & 0xffff) / 65536.0f;
  };
  for (auto& v : hg) v = 0.05f + rnd();
  for (auto& v : hh) v = 0.05f + rnd();
  for (auto& v : hx) v = (float)((int)(rnd() * 4.0f));   // counts 0..3
  float *g, *h, *x, *out;
  CHECK(hipMalloc(&g, hg.size() * 4));
  CHECK(hipMalloc(&h, hh.size() * 4));
  CHECK(hipMalloc(&x, hx.size() * 4));
  CHECK(hipMalloc(&out, hg.size() * 4));
  CHECK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> first, cur(hg.size());
  printf("%s, %d CUs; NOPS=%d HOLD=%d CHAINS=%d; %d workgroups of 256 threads, %d steps, %d launches per line\n", prop.gcnArchName, prop.multiProcessorCount,
         NOPS, HOLD, CHAINS, blocks, steps, reps);
  // 100 KB of LDS per workgroup: one workgroup (4 waves = one wave per SIMD) per CU; 32 KB: two or more (two waves per SIMD and up)
  run<16>("16-slot form, one wave per SIMD", 100 * 1024, blocks, steps, reps, g, h, x, out, first, cur);
  std::vector<float> ref16 = first;
  run<16>("16-slot form, shared SIMDs", 32 * 1024, blocks, steps, reps, g, h, x, out, first, cur);
  run<32>("32-slot form, one wave per SIMD", 100 * 1024, blocks, steps, reps, g, h, x, out, first, cur);
  std::vector<float> ref32 = first;
  run<32>("32-slot form, shared SIMDs", 32 * 1024, blocks, steps, reps, g, h, x, out, first, cur);
  // the shared-SIMD result of the 32-slot form against its own one-wave-per-SIMD result (the same arithmetic: equal bits when nothing races)
  long d = 0;
  for (size_t i = 0; i < first.size(); ++i) d += memcmp(&first[i], &ref32[i], 4) != 0;
  printf("32-slot form: first shared-SIMD launch against the one-wave-per-SIMD result: %ld of %zu values differ\n", d, first.size());
  double m = 0;
  for (size_t i = 0; i < ref16.size(); ++i) m = fmax(m, fabs((double)ref16[i] - ref32[i]) / fmax(fabs((double)ref16[i]), 1e-30));
  printf("16-slot against 32-slot form (another order of the partial products): worst relative difference %.2e\n", m);
  return 0;
}
