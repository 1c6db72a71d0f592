// Dense H-step on the matrix cores: 9..16 components (the wide build, KP = 16), 17..32 in two halves of 16 (the widest, KP = 32).
//
// updates.py:127-132 on a dense store: Y = GW H, R = X / Y, num = GW^T R - the (GW)^T X contraction BASELINE.json's north
// star names - plus the KL term of the input state.  Same reasoning as mu_w_mfma_kernel.hpp: from 9 components on the two
// contractions are most of the vector kernel's 2 k + 6 instructions per element and a 16-wide tile is 56..100 % full; both
// run here as v_mfma_f32_16x16x16_bf16 on operands split into bf16 hi + lo (three products, fp32-grade sums), and the vector
// ALU keeps the count -> float conversion, the reciprocal, the ratio, the loss term and the split of R.
//
// One workgroup = 4 waves = a tile of TP = 16 STEPS PASSES pixels (256 or 128), worked in PASSES passes of 16 STEPS = 128
// pixels (the H operands and accumulators of 8 steps fit the registers next to the GW operands; 16 spilled); the waves
// split the channel range in tiles of 64 channels and each leaves a partial numerator in LDS, which the shared per-pixel
// epilogue (h_epilogue) sums.
// Per (channel tile T, sub-tile j = 0..3, pixel step s):
//   1. Y (16 channels x 16 pixels) = GW (channels x k) . H (k x pixels)
//        A operand: lane l holds GW[channel of row l % 16, components 4 (l / 16) .. + 3]      (16 bytes of gw_s)
//        B operand: lane l holds H[components 4 (l / 16) + i, pixel 16 s + l % 16]             (registers, constant over the channels)
//        result   : lane l holds Y[rows 4 (l / 16) + r, pixel 16 s + l % 16]
//      Row m = 4 q + r of sub-tile j is channel 64 T + 16 q + 4 j + r: the 4 rows of a lane in the four sub-tiles are 16
//      CONSECUTIVE channels of its pixel, i.e. ONE 16-byte load of the pixel-major X (8-bit store) serves all four.
//   2. R = X / Y in that layout; KL term x log2(x / y)
//   3. num^T (16 pixels x 16 components) += R^T (pixels x channels) . GW (channels x k)
//        A operand: lane l holds R[channels of rows 4 (l / 16) + i, pixel l % 16]  - the result layout of step 1
//        B operand: lane l holds GW[channel of row 4 (l / 16) + i, component l % 16]
//        result   : lane l holds num[component l % 16, pixels 16 s + 4 (l / 16) + r]  -> one 16-byte LDS store per step
#pragma once
#include "mu_h_kernel.hpp"
#include "mu_w_mfma_kernel.hpp"

// pixel steps of 16 per pass, workgroups per CU the register budget is set for (A/B: tools/analysis/build_variant_lib.sh)
#ifndef ESPM_H_MFMA_STEPS
#define ESPM_H_MFMA_STEPS 4
#endif
#ifndef ESPM_H_MFMA_MINBLK
#define ESPM_H_MFMA_MINBLK 2
#endif

namespace espm {

// 16 consecutive channels of one pixel row of the pixel-major X (p, n_pad); dwords beyond n_pad (the last channel tile)
// and pixels beyond p read as zero counts
template <typename XT>
__device__ __forceinline__ void mf_load_px_row(MfRow<XT>& row, const XT* x_pm, int px, int p, int c0, int n_pad) {
  constexpr int EPD = 4 / (int)sizeof(XT) > 0 ? 4 / (int)sizeof(XT) : 1;   // elements per dword: 4 (u8), 2 (bf16), 1 (f32)
  constexpr int NDW = 16 / EPD;                                            // dwords of 16 channels
  uint32_t* w = reinterpret_cast<uint32_t*>(&row);
  static_assert(sizeof(MfRow<XT>) == NDW * 4, "MfRow is the 16 channels, nothing else");
  // (16-byte loads need rows that start on 16 bytes: always with 2- and 4-byte elements, with bytes when n_pad is a multiple of 16)
  if (px < p && c0 + 16 <= n_pad && ((size_t)n_pad * sizeof(XT)) % 16 == 0) {
    row.load(x_pm + (size_t)px * n_pad + c0);
    return;
  }
  const uint32_t* src = reinterpret_cast<const uint32_t*>(x_pm + (size_t)min(px, p - 1) * n_pad);
#pragma unroll
  for (int d = 0; d < NDW; ++d) {
    const int c = c0 + d * EPD;
    w[d] = (px < p && c < n_pad) ? src[c / EPD] : 0u;
  }
}

template <int K, typename XT, int STEPS, int PASSES, bool LOSS>
__global__ __launch_bounds__(256, ESPM_H_MFMA_MINBLK) void h_step_mfma_kernel(const HStepArgs a) {   // (two workgroups per CU by their LDS: <= 256 registers)
  static_assert(KP == 32 || KP == 16 || KP == 8, "component stride 8, 16 or 32: a 16-wide tile is zero-filled beyond it");
  constexpr int TP = 16 * STEPS * PASSES, NW = 4;
  constexpr int KH = (K + 15) / 16;   // halves of 16 components (mu_w_mfma_kernel.hpp: two from 17 components on)
  static_assert(KH <= MF_KH, "k beyond the stride");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][K][TP]
  const int lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * TP;
  const XT* x_pm = static_cast<const XT*>(a.x_pm);
  const int tiles = (a.n_pad + 63) / 64;
  float kl = 0.f;
  // GW operands of one channel tile, both layouts, four sub-tiles: requested one tile ahead of their use
  struct GRaw {
    float4 a[4][KH];     // sub-tile j: GW[channel of row l16, components 16 hf + 4 q .. + 3]
    float b[4][KH][4];   // sub-tile j: GW[channel of row 4 q + i, component 16 hf + l16]
  };
  auto load_g = [&](GRaw& g, int T) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ca = min(64 * T + 16 * (l16 >> 2) + 4 * j + (l16 & 3), a.n_pad - 1);   // (rows beyond n_pad: X = 0 there)
#pragma unroll
      for (int hf = 0; hf < KH; ++hf) {
        g.a[j][hf] = (4 * q < MF_KW) ? *reinterpret_cast<const float4*>(a.gw_s + (size_t)ca * KP + 16 * hf + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) g.b[j][hf][i] = l16 < MF_KW ? a.gw_s[(size_t)min(64 * T + 16 * q + 4 * j + i, a.n_pad - 1) * KP + 16 * hf + l16] : 0.f;
      }
    }
  };

#pragma unroll 1
  for (int ps = 0; ps < PASSES; ++ps) {
    const int px0 = tile0 + ps * 16 * STEPS;
    // B operand of step 1: H[components 16 hf + 4 q + i, pixel 16 s + l16], split once (pad pixels hold positive values)
    mf_s4 hh[STEPS][KH], hl[STEPS][KH];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int hf = 0; hf < KH; ++hf) {
        float hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = (16 * hf + 4 * q + i < K) ? a.h_in[(size_t)(16 * hf + 4 * q + i) * a.p_pad + px0 + 16 * s + l16] : 0.f;
        mf_split(hv, hh[s][hf], hl[s][hf]);
      }
    }
    mf_f4 acc[STEPS][KH];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int hf = 0; hf < KH; ++hf) acc[s][hf] = mf_f4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int T = wave; T < tiles; T += NW) {
      GRaw gn;
      load_g(gn, T);
      mf_s4 a1h[4][KH], a1l[4][KH], b3h[4][KH], b3l[4][KH];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int hf = 0; hf < KH; ++hf) {
          const float gav[4] = {gn.a[j][hf].x, gn.a[j][hf].y, gn.a[j][hf].z, gn.a[j][hf].w};
          mf_split(gav, a1h[j][hf], a1l[j][hf]);
          mf_split(gn.b[j][hf], b3h[j][hf], b3l[j][hf]);
        }
      }
      const int c0 = 64 * T + 16 * q;
      MfRow<XT> xr[2];
      mf_load_px_row<XT>(xr[0], x_pm, px0 + l16, a.p, c0, a.n_pad);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (s + 1 < STEPS) mf_load_px_row<XT>(xr[(s + 1) & 1], x_pm, px0 + 16 * (s + 1) + l16, a.p, c0, a.n_pad);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x[4], r[4];
          xr[s & 1].quad(j, x);
          mf_f4 y = mf_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int hf = KH - 1; hf >= 0; --hf) y = mf_mma3<1>(a1h[j][hf], a1l[j][hf], hh[s][hf], hl[s][hf], y);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // The first reader of a matrix-core result must be an instruction the compiler knows (it owes the wait states
            // between v_mfma and a vector read of its result; it cannot see into the asm below and places none for it).
            float yi = fmaxf(y[i], 1e-37f);
            asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0" : "+v"(yi));   // in place: the transcendental unit reads its source late (the matrix-core hazard); s_nop: the wait state a vector instruction that reads a transcendental result needs - the compiler's hazard recognizer does not see into the asm
            // (+1e-37 with the loss: keeps log2(R) finite where X = 0, as in the vector kernels)
            r[i] = LOSS ? fmaf(x[i], yi, 1e-37f) : x[i] * yi;
            if constexpr (LOSS) {
              float lg = r[i];
              asm volatile("v_log_f32 %0, %0\n\ts_nop 0" : "+v"(lg));   // in place, like the reciprocal (and with its wait state)
              kl = fmaf(x[i], lg, kl);
            }
          }
          mf_s4 rh, rl;
          mf_split(r, rh, rl);
#pragma unroll
          for (int hf = 0; hf < KH; ++hf) acc[s][hf] = mf_mma3<2>(rh, rl, b3h[j][hf], b3l[j][hf], acc[s][hf]);
        }
        // the four sub-tile chains of a step are enough to keep the matrix and the vector pipes busy; without the fence the
        // scheduler interleaves all eight steps and spills 250 registers (with the loss term)
#ifndef ESPM_H_MFMA_NO_FENCE   // (diagnosis of the 32-slot form's run-to-run differences)
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    // partial numerators of this wave: num[component 16 hf + l16, pixels 16 s + 4 q + r of the pass]
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) {
      if (16 * hf + l16 < K) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
          *reinterpret_cast<float4*>(smem + ((size_t)wave * K + 16 * hf + l16) * TP + ps * 16 * STEPS + 16 * s + 4 * q) =
              make_float4(acc[s][hf][0], acc[s][hf][1], acc[s][hf][2], acc[s][hf][3]);
      }
    }
  }
  h_epilogue<K, true, 0>(a, smem, NW, TP, tile0, LOSS ? kl : 0.f);
}

}  // namespace espm
