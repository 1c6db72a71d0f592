// Micro-benchmark (gfx950): does the matrix pipe take work off the vector pipe "for free" in the dense H-step's inner loop?
//
// VERDICT r4 item 4(a).  The dense kernels (h_step_kernel / w_accum_kernel, mu_h_kernel.hpp / mu_w_kernel.hpp) spend, per pixel
// pair and channel at k = 5, 5 + 5 packed FMAs on the two contractions and ~7-12 issue slots on the element-wise stream between
// them (count -> float, reciprocal, ratio, every eighth logarithm).  The question is whether the contractions, moved to matrix
// instructions issued FROM THE SAME WAVES, overlap that stream: the vector and the matrix pipe of a SIMD are separate units, but
// a wave issues in order, so the overlap has to come from the other waves of the SIMD.
//
// Every kernel runs ITER iterations of a loop body per wave, 4 waves per SIMD on all 1024 SIMDs (one workgroup of 256 threads per
// SIMD-quad and wave slot):
//   ew        the element-wise stream of 16 elements per lane: v_cvt_f32_ubyte0-3, v_rcp_f32, v_mul_f32, 2 x v_log_f32, 2 x v_fma_f32
//   ew+fma    ew + the two contractions as the vector kernel does them: 16 x (5 + 5) FMAs in fp32x2 registers (80 v_pk_fma_f32)
//   ew+mfma32 ew + NM x v_mfma_f32_16x16x4_f32  (exact fp32, the vector pipe's rate: 256 flop / cycle / CU)
//   ew+mfma16 ew + NM x v_mfma_f32_32x32x16_bf16 (hi / lo split operands; 16 x the rate)
//   mfma32 / mfma16 alone
// NM is chosen so that the matrix instructions cover the same 16 elements x 64 lanes of both contractions:
//   fp32 16x16x4:    a 16 x 16 tile of Y with depth 8 (k = 5 padded) = 2 instructions per 256 values; the second contraction, 16 components
//                    x 16 pixels with depth 16 channels = 4 per 256 values: 6 per 256 values -> 24 per body (1024 values)
//   bf16 32x32x16:   Y: depth 15 of 16 (hi hi, hi lo, lo hi at k = 5) = 1 per 1024 values; second: depth 32 channels x 3 products = 6 per
//                    1024 values: 7 per body
// The printed figure is cycles per body per SIMD (4 waves round-robin) at the measured clock (wall clock / s_memtime ratio is not
// needed: all modes are compared with each other at the same clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));

constexpr int ITER = 2048;

template <bool EW, bool FMA, int NM32, int NM16, bool SPLIT = false>
__global__ __launch_bounds__(256) void body(float* out, const unsigned* in, float seed) {
  // element-wise state: 4 packed words of 4 counts each = 16 elements per lane
  unsigned w[4];
  for (int i = 0; i < 4; ++i) w[i] = in[(threadIdx.x + i * 256) & 1023] | 0x01010101u;
  float y[16], kl = 0.f;
  for (int i = 0; i < 16; ++i) y[i] = seed + 0.5f + i * 0.01f + threadIdx.x * 1e-4f;
  f2 h[5], g[5], num[5];
  for (int i = 0; i < 5; ++i) {
    h[i] = f2{seed + i, seed - i};
    g[i] = f2{0.5f + i, 0.25f * i + 0.125f};
    num[i] = f2{0.f, 0.f};
  }
  f4 acc32[6];
  for (int i = 0; i < 6; ++i) acc32[i] = f4{0.f, 0.f, 0.f, 0.f};
  f16v acc16[2];
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 16; ++i) acc16[j][i] = 0.f;
  float a32 = seed * 0.001f + threadIdx.x * 1e-6f, b32 = seed * 0.002f;
  s8 a16, b16;
  for (int i = 0; i < 8; ++i) {
    a16[i] = (short)(0x3c00 + i);
    b16[i] = (short)(0x3c10 + i + (threadIdx.x & 7));
  }
  for (int it = 0; it < ITER; ++it) {
    if (EW) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x0, x1, x2, x3;
        asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x0) : "v"(w[q]));
        asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x1) : "v"(w[q]));
        asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x2) : "v"(w[q]));
        asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(x3) : "v"(w[q]));
        float x[4] = {x0, x1, x2, x3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float r = __builtin_amdgcn_rcpf(y[4 * q + i]);
          r = r * x[i];
          y[4 * q + i] = fmaf(r, 1e-3f, 1.0f);   // keeps y bounded and dependent
        }
      }
      // two logarithms per 16 elements (the product of eight ratios each)
      kl += __builtin_amdgcn_logf(y[0] * y[5]) + __builtin_amdgcn_logf(y[9] * y[14]);
    }
    if (FMA) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // 8 pixel pairs = 16 elements: 5 + 5 packed FMAs each
        asm volatile("" : "+v"(h[0]));   // (another pixel pair's H every time: the dot product is not loop-invariant)
        f2 yy = g[0] * h[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) yy = g[k] * h[k] + yy;
#pragma unroll
        for (int k = 0; k < 5; ++k) num[k] = g[k] * yy + num[k];
      }
    }
    if (SPLIT) {   // the ratios as bf16 hi + lo pairs, the matrix instruction's operand form: and, subtract, two byte permutes per two elements
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const unsigned r0 = __float_as_uint(y[i]), r1 = __float_as_uint(y[i + 1]);
        const unsigned h0 = r0 & 0xffff0000u, h1 = r1 & 0xffff0000u;
        const float l0 = y[i] - __uint_as_float(h0), l1 = y[i + 1] - __uint_as_float(h1);
        const unsigned ph = __builtin_amdgcn_perm(h1, h0, 0x07060302u), pl = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        a16[(i >> 1) & 7] = (short)(ph ^ pl);
        b16[(i >> 1) & 7] = (short)((ph >> 16) ^ (pl >> 16));
      }
    }
    if (NM32 > 0) {
#pragma unroll
      for (int i = 0; i < NM32; ++i) acc32[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc32[i % 6], 0, 0, 0);
    }
    if (NM16 > 0) {
#pragma unroll
      for (int i = 0; i < NM16; ++i) acc16[i % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc16[i % 2], 0, 0, 0);
    }
  }
  float s = kl;
  for (int i = 0; i < 16; ++i) s += y[i];
  for (int i = 0; i < 5; ++i) s += num[i].x + num[i].y;
  for (int i = 0; i < 6; ++i) s += acc32[i][0] + acc32[i][3];
  for (int j = 0; j < 2; ++j) s += acc16[j][0] + acc16[j][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool EW, bool FMA, int NM32, int NM16, bool SPLIT = false>
double run(const char* name, float* out, const unsigned* in, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((body<EW, FMA, NM32, NM16, SPLIT>), dim3(blocks), dim3(256), 0, 0, out, in, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((body<EW, FMA, NM32, NM16, SPLIT>), dim3(blocks), dim3(256), 0, 0, out, in, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double ns_per_body = best * 1e6 / ((double)ITER * waves_per_simd);   // per SIMD: its waves' bodies run one after another
  printf("%-34s waves/SIMD %d: %8.3f ms -> %7.1f ns per body per SIMD (%6.1f cycles @2.4 GHz)\n", name, waves_per_simd, best, ns_per_body, ns_per_body * 2.4);
  return ns_per_body;
}

int main() {
  float* out;
  unsigned* in;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&in, 1024 * sizeof(unsigned));
  hipMemset(in, 3, 1024 * sizeof(unsigned));
  for (int w : {4, 2}) {
    const double ew = run<true, false, 0, 0>("ew (element-wise stream)", out, in, w);
    const double fma = run<false, true, 0, 0>("fma (80 v_pk_fma_f32)", out, in, w);
    const double ewfma = run<true, true, 0, 0>("ew + fma (the vector kernel)", out, in, w);
    const double m32 = run<false, false, 24, 0>("mfma32 (24 x 16x16x4 f32)", out, in, w);
    const double ewm32 = run<true, false, 24, 0>("ew + mfma32", out, in, w);
    const double m16 = run<false, false, 0, 7>("mfma16 (7 x 32x32x16 bf16)", out, in, w);
    const double ewm16 = run<true, false, 0, 7>("ew + mfma16", out, in, w);
    const double ewm16s = run<true, false, 0, 7, true>("ew + split + mfma16", out, in, w);
    printf("  overlap: ew + mfma32 = %.2f x max(ew, mfma32), %.2f x (ew + mfma32 serial); ew + mfma16 = %.2f x max, %.2f x serial; the vector kernel = %.2f x (ew + fma serial)\n",
           ewm32 / (ew > m32 ? ew : m32), ewm32 / (ew + m32), ewm16 / (ew > m16 ? ew : m16), ewm16 / (ew + m16), ewfma / (ew + fma));
    printf("  against the vector kernel's body: matrix fp32 %.2f x, matrix bf16 hi/lo %.2f x without / %.2f x WITH the split of the ratios into bf16 hi + lo (the operands the matrix instruction needs)\n",
           ewfma / ewm32, ewfma / ewm16, ewfma / ewm16s);
  }
  return 0;
}
