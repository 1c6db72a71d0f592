// Micro-benchmark: issue cost of the VALU instructions the MU kernels are made of (gfx950).
// Each kernel runs ITER iterations of 16 independent instructions of one kind per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[16];
  float2_ p[8];
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
  for (int i = 0; i < 8; ++i) p[i] = float2_{a[2 * i], a[2 * i + 1]};
  float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.001f;
  float2_ pb = {b, b}, pc = {c, c};
  for (int it = 0; it < ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 5) {  // 12 fma + 4 rcp interleaved (can the transcendental unit overlap?)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a[4 * i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4 * i + 1]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4 * i + 2]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4 * i + 3]) : "v"(b), "v"(c));
      }
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
    } else if (MODE == 7) {  // fma with an SGPR operand
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(c));
    } else if (MODE == 8) {  // pk_fma with a broadcast SGPR pair operand (low half for both lanes)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(pb), "v"(pc));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(pb), "v"(pc));
    } else if (MODE == 10) {  // mixed-precision fma: f16 (low half) * f32 + f32
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 11) {  // v_perm_b32 (bf16 -> f32 widening by byte select)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 12) {  // v_and_b32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a[i]));
    } else if (MODE == 13) {  // v_mov_b32 from sgpr
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(seed));
    } else if (MODE == 14) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_u32_u24 %0, 0x10000, %0" : "+v"(a[i]));
    } else if (MODE == 15) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshl_add_u32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 17) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
    } else if (MODE == 18) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_i32_i24 %0, 0x10000, %0" : "+v"(a[i]));
    } else if (MODE == 19) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 9) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* out, double flops_per_instr) {
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr_per_simd = (double)ITER * 16 * waves_per_simd;  // wave-instructions per SIMD
  const double ghz = 2.4;
  printf("%-28s waves/SIMD %d: %8.3f ms  -> %6.2f cycles per wave-instruction @%.1f GHz (%.1f TFLOP/s-equivalent)\n", name,
         waves_per_simd, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz,
         flops_per_instr * 64 * instr_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {4}) {
    run<0>("v_fma_f32", w, out, 2);
    run<1>("v_pk_fma_f32", w, out, 4);
    run<4>("v_mul_f32", w, out, 1);
    run<9>("v_pk_mul_f32", w, out, 2);
    run<2>("v_rcp_f32", w, out, 1);
    run<3>("v_log_f32", w, out, 1);
    run<5>("4 rcp + 12 fma", w, out, 1.75);
    run<6>("v_lshlrev_b32", w, out, 1);
    run<7>("v_fma_f32 sgpr", w, out, 2);
    run<8>("v_pk_fma_f32 sgpr bcast", w, out, 4);
    run<10>("v_fma_mix_f32 (f16*f32+f32)", w, out, 2);
    run<11>("v_perm_b32", w, out, 1);
    run<12>("v_and_b32 imm", w, out, 1);
    run<13>("v_mov_b32 sgpr", w, out, 1);
    run<14>("v_mul_u32_u24", w, out, 1);
    run<15>("v_alignbit_b32", w, out, 1);
    run<16>("v_lshl_add_u32", w, out, 1);
    run<17>("v_cvt_f32_f16", w, out, 1);
    run<18>("v_mul_i32_i24", w, out, 1);
    run<19>("v_max_f32", w, out, 1);
  }
  return 0;
}
