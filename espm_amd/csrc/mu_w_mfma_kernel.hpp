// Dense W accumulation on the matrix cores: 9..16 components (the wide build, KP = 16), 17..32 in two halves of 16 (the widest build,
// KP = 32) and, with half-empty tiles, 7..8.
//
// updates.py:38-39, :53, :59 on a dense store: Y = GW H', R = X / Y, A = R H'^T.  With k <= 8 the two contractions are
// 2 k of the ~2 k + 6 vector instructions an element costs, and padding k to a matrix-core tile wastes most of the tile:
// the matrix-core H-step measured slower at k = 5.  From k = 9 on the balance tips: the wide build's vector
// kernels spend 2 k + 6 = 24..38 instructions per element, while a 16-wide tile is 56..100 % full.  Both contractions of
// the W accumulation run here as v_mfma_f32_16x16x16_bf16 with every fp32 operand split into bf16 hi + lo (three products:
// hi hi + hi lo + lo hi, relative error ~2^-16 per product, random sign: the sums over 10^5 pixels are fp32-grade), and
// what stays on the vector ALU per element is the count -> float conversion, one reciprocal, one multiply and the split of
// R (~6 instructions).
//
// Tiling (one wave): CT = 8 channel tiles of 16 channels against groups of 64 pixels = 4 steps of 16 pixel slots.
//   step s, slot m = 4 q + r  <->  pixel 16 q + 4 s + r of the group (so that a lane's 16 pixels of a channel row are ONE
//   16-byte load of the tile-major X for all four steps).
//   1. Y^T (16 pixel slots x 16 channels) = H'^T (slots x k) . GW^T (k x channels):
//        A operand: lane l holds H'[4 (l / 16) .. + 3, pixel of slot l % 16]   (16 bytes of h_t)
//        B operand: lane l holds GW[channel l % 16, 4 (l / 16) .. + 3]          (registers, constant over the pixels)
//        result   : lane l holds Y[channel l % 16, slots 4 (l / 16) + r]       - the A-operand layout of step 3
//   2. R = X / Y in that layout (X: the lane's channel row, pixels of slots 4 (l / 16) + r: one dword of its 16-byte load)
//   3. A (16 channels x 16 components) += R (channels x slots) . H'^T (slots x k):
//        B operand: lane l holds H'[component l % 16, pixels of slots 4 (l / 16) + r]   (four dwords of h_t)
//        result   : lane l holds A[channels 4 (l / 16) + r, component l % 16]  -> one 16-byte store into the slab
// The H' operands of a pixel group are loaded and split once and serve all eight channel tiles.
#pragma once
#include "mu_common.hpp"

namespace espm {

typedef short mf_s4 __attribute__((ext_vector_type(4)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));
typedef float mf_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 mf_b2 __attribute__((ext_vector_type(2)));
typedef unsigned mf_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mf_pack(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(mf_f2{a, b}, mf_b2));
}
// v = hi + lo + O(2^-17 |v|) with hi, lo bf16
__device__ __forceinline__ void mf_split(const float (&v)[4], mf_s4& hi, mf_s4& lo) {
  const unsigned h0 = mf_pack(v[0], v[1]), h1 = mf_pack(v[2], v[3]);
  const unsigned l0 = mf_pack(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
  const unsigned l1 = mf_pack(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
  hi = __builtin_bit_cast(mf_s4, mf_u2{h0, h1});
  lo = __builtin_bit_cast(mf_s4, mf_u2{l0, l1});
}
// c += (ah + al) . (bh + bl) over the 16 slots of the contraction.
// The 32-slot form (gfx950's own shape): v_mfma_f32_16x16x32_bf16 contracts over 32 slots in the cycles the CDNA3-era 16x16x16 form
// takes for 16, and the hi / lo split supplies exactly two 16-slot halves: with the A operand [ah | al] (the two halves side by
// side in one 8-element operand - slot i of either half is the same component, so the order inside the 32 does not matter as
// long as both operands agree) one instruction against [bl | bl] gives ah bl + al bl and one against [bh | bh] gives
// ah bh + al bh: the three products of the 16-wide form (and the fourth, lo lo, for free) in TWO matrix instructions instead of
// three.  What it costs is the duplicated operand: four registers instead of two, filled by two moves where the half is not
// already held twice.
typedef short mf_s8 __attribute__((ext_vector_type(8)));
typedef __bf16 mf_b8 __attribute__((ext_vector_type(8)));
// Where it is used (ESPM_MFMA_K32_MASK, one bit per call site - 1: Y of the H-step, 2: the H-step's numerator, 4: Y of the W
// accumulation, 8: the W accumulation's sums; default 12 = the W accumulation, both sites):
//  * W accumulation: bit-reproducible from run to run at k = 9, 12, 16 and 0.6-1.4 % of the iteration faster than the 16-slot form
//    (k = 16: 662 against 667 us, k = 12: 632.5 against 641.5; profiles/r03z_wide_repro_*.log).
//  * H-step: NOT reproducible with either of its sites alone (mask 1, mask 2: 2-3 million of the 4 million entries of H differ in
//    their last bits - max 3e-5 - between two runs from the same state), with every variant tried: operands held across 1, 2, 4, 8
//    wait states after the issue (ESPM_MFMA_K32_NOPS), 20 wait states behind the second instruction (256 entries still differ:
//    delay helps, so it is an ordering hazard, not arithmetic), profiles/r03d_wide_repro.log, r03e_wide_repro.log, r03z_wide_repro_m{1,2}.log.
//    In the H-step kernel it is a matter of how many waves share a SIMD (245 registers, two workgroups per CU: TWO): built for one
//    (ESPM_H_MFMA_MINBLK=1) it is bit-reproducible
//    with the 32-slot form on all four sites too, only slower than everything else (792 us; profiles/r03ab_wide_repro_h3one.log;
//    without its scheduling fence it still differs, r03ab_wide_repro_h3nofence.log).  The clean form of that experiment: the SAME
//    machine code, launched with 24 KB of LDS nobody uses so that one workgroup fits a CU instead of two (ESPM_H_MFMA_PAD_LDS,
//    mu_h_step.hip) - bit-reproducible, twice; launched normally - 2.7-3.1 million entries differ, twice
//    (profiles/r03an_wide_repro_h3{pad,two}.log).  Occupancy, not code.  (Keeping the small terms in a register set of their own,
//    so that no matrix instruction takes the result of the one before it, made differences rarer - one run in six - and the
//    iteration 20 % slower: r03ao_*.)  Two waves per SIMD are not SUFFICIENT, though: the W kernel on the 8-bit store takes 256
//    registers (36 of them accumulation registers) - two waves per SIMD as well - and has not shown one differing bit in some thirty
//    runs of three iterations (masks 4, 8, 12, the product; k = 9, 12, 16); forced down to one wave per SIMD (LDS reserved at the
//    launch, or amdgpu_waves_per_eu(1, 1)) it loses 9 % (724 against 662 us, profiles/r03ap_*) and stays as it was.  What the
//    H-step kernel has on top: LDS traffic, 183 spilled scalar registers (lane writes / reads), the logarithm.  So: with two waves interleaving 32-slot matrix
//    instructions on one SIMD, either a result is read (site 1: by the vector ALU eight wait states later, the compiler's count)
//    or an operand is rewritten (site 2) before the matrix pipe - busy with the other wave's instruction - has got to it; the
//    16-slot form does not show it.  A hardware interlock the new shape lacks or a wait-state table this compiler has too short:
//    not decided here.  All four sites together would be 6.5 % at k = 16 (630 against 674 us); a result that changes from run to
//    run is not shipped for that.
#ifndef ESPM_MFMA_K32_MASK
#define ESPM_MFMA_K32_MASK 12
#endif
// (Diagnosis knob: the empty asm keeps both four-register operands allocated across ESPM_MFMA_K32_NOPS + 1 wait states after the
// matrix instruction - hipcc reuses one of them two or three instructions later.  It did not restore reproducibility in the H-step.)
#ifdef ESPM_MFMA_K32_NOPS
#define ESPM_STR2(x) #x
#define ESPM_STR(x) ESPM_STR2(x)
#define ESPM_MFMA_K32_HOLD(a, b) asm volatile("s_nop " ESPM_STR(ESPM_MFMA_K32_NOPS) : : "v"(a), "v"(b))
#else
#define ESPM_MFMA_K32_HOLD(a, b)
#endif
template <int SITE>
__device__ __forceinline__ mf_f4 mf_mma3(const mf_s4 ah, const mf_s4 al, const mf_s4 bh, const mf_s4 bl, mf_f4 c) {
  if constexpr ((ESPM_MFMA_K32_MASK & SITE) != 0) {
  const mf_b8 a = __builtin_bit_cast(mf_b8, __builtin_shufflevector(ah, al, 0, 1, 2, 3, 4, 5, 6, 7));
  const mf_b8 b1 = __builtin_bit_cast(mf_b8, __builtin_shufflevector(bl, bl, 0, 1, 2, 3, 4, 5, 6, 7));
  const mf_b8 b2 = __builtin_bit_cast(mf_b8, __builtin_shufflevector(bh, bh, 0, 1, 2, 3, 4, 5, 6, 7));
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, c, 0, 0, 0);   // small terms first
  ESPM_MFMA_K32_HOLD(a, b1);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c, 0, 0, 0);
  ESPM_MFMA_K32_HOLD(a, b2);
  return c;
  } else {
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);   // small terms first
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
  }
}

// 16 consecutive pixels of one channel row of the tile-major X; quad(s) = the 4 pixels of step s as floats
template <typename XT>
struct MfRow;
template <>
struct MfRow<uint8_t> {
  uint4 v;
  __device__ __forceinline__ void load(const uint8_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void quad(int s, float (&x)[4]) const {
    const uint32_t w = s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
    x[0] = ub0(w); x[1] = ub1(w); x[2] = ub2(w); x[3] = ub3(w);
  }
};
template <>
struct MfRow<bf16_t> {
  uint4 v[2];
  __device__ __forceinline__ void load(const bf16_t* p) {
    v[0] = reinterpret_cast<const uint4*>(p)[0];
    v[1] = reinterpret_cast<const uint4*>(p)[1];
  }
  __device__ __forceinline__ void quad(int s, float (&x)[4]) const {
    const uint4 q = v[s >> 1];
    const uint32_t a = (s & 1) ? q.z : q.x, b = (s & 1) ? q.w : q.y;
    x[0] = __uint_as_float(a << 16); x[1] = __uint_as_float(a & 0xffff0000u);
    x[2] = __uint_as_float(b << 16); x[3] = __uint_as_float(b & 0xffff0000u);
  }
};
template <>
struct MfRow<float> {
  float4 v[4];
  __device__ __forceinline__ void load(const float* p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = reinterpret_cast<const float4*>(p)[i];
  }
  __device__ __forceinline__ void quad(int s, float (&x)[4]) const {
    const float4 q = v[s];
    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
  }
};

// (The Frobenius variant - L2: no Y product - keeps the 16-slot form: it was not part of the reproducibility runs above.)
// 17..32 components (the widest build, KP = 32): the component range in MF_KH = 2 halves of 16 - step 1 contracts over both (two chains into
// the same Y), step 3 keeps one accumulator tile per half; MF_CT channel tiles per wave (8, like the 16-component build: with 4 the
// H' operands of a pixel group were loaded and split for half as many channels).
constexpr int MF_KH = KP > 16 ? KP / 16 : 1;
#ifndef ESPM_MF_CT32
#define ESPM_MF_CT32 8   // channel tiles per wave with two halves of components (A/B with tools/analysis/build_variant_wide32.sh, profiles/r05z_wide32_ab.log: 8-bit store 1038 -> 928 us per iteration at k = 17, 1321 -> 1208 at 32 against 4 tiles; bf16 store unchanged)
#endif
constexpr int MF_CT = KP > 16 ? ESPM_MF_CT32 : 8;
constexpr int MF_KW = KP < 16 ? KP : 16;   // components of one half that exist in the KP-strided rows

template <int K, typename XT, bool L2>
__device__ __forceinline__ void w_accum_mfma_body(const WAccumArgs& a) {
  static_assert(KP == 32 || KP == 16 || KP == 8, "component stride 8, 16 or 32: a 16-wide tile is zero-filled beyond it");
  constexpr int CT = MF_CT;                      // channel tiles of 16 per wave
  constexpr int KH = (K + 15) / 16;              // halves that hold components
  static_assert(KH <= MF_KH, "k beyond the stride");
  const int lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cbase = (blockIdx.y * 4 + wave) * 16 * CT;
  if (cbase >= a.n_pad) return;                  // (whole waves)

  // B operand of step 1: GW[channel, 16 hf + 4 q .. + 3] per channel tile, split once
  mf_s4 gh[CT][KH], gl[CT][KH];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = min(cbase + 16 * t + l16, a.n_pad - 1);
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (4 * q < MF_KW) g = *reinterpret_cast<const float4*>(a.gw_s + (size_t)c * KP + 16 * hf + 4 * q);
      const float gv[4] = {g.x, g.y, g.z, g.w};
      mf_split(gv, gh[t][hf], gl[t][hf]);
    }
  }
  mf_f4 acc[CT][KH];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) acc[t][hf] = mf_f4{0.f, 0.f, 0.f, 0.f};

  const int groups = a.p_pad / 64;
  const int gpb = (groups + (int)gridDim.x - 1) / (int)gridDim.x;
  const int g_begin = blockIdx.x * gpb, g_end = min(groups, g_begin + gpb);
  const XT* x_cm = static_cast<const XT*>(a.x_cm);
  // raw H' operands of a pixel group (both layouts, four steps): requested one group ahead of their use
  struct HRaw {
    float4 a[4][KH];      // step s: H'[16 hf + 4 q .. + 3, pixel of slot l16]
    float b[4][KH][4];    // step s: H'[component 16 hf + l16, pixels of slots 4 q + r]
  };
  auto load_h = [&](HRaw& h, int g) {
    const int px0 = g * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int pa = min(px0 + 16 * (l16 >> 2) + 4 * s + (l16 & 3), a.p - 1);      // pixel of slot l16 (rows beyond p: X = 0)
#pragma unroll
      for (int hf = 0; hf < KH; ++hf) {
        h.a[s][hf] = (4 * q < MF_KW) ? *reinterpret_cast<const float4*>(a.h_t + (size_t)pa * KP + 16 * hf + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) h.b[s][hf][r] = l16 < MF_KW ? a.h_t[(size_t)min(px0 + 16 * q + 4 * s + r, a.p - 1) * KP + 16 * hf + l16] : 0.f;
      }
    }
  };
  HRaw hn;
  if (g_begin < g_end) load_h(hn, g_begin);
  for (int g = g_begin; g < g_end; ++g) {
    const int px0 = g * 64;
    if (px0 >= a.p) break;                       // (only padding beyond: X = 0 there)
    mf_s4 a1h[4][KH], a1l[4][KH], b3h[4][KH], b3l[4][KH];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int hf = 0; hf < KH; ++hf) {
        const float hav[4] = {hn.a[s][hf].x, hn.a[s][hf].y, hn.a[s][hf].z, hn.a[s][hf].w};
        mf_split(hav, a1h[s][hf], a1l[s][hf]);
        mf_split(hn.b[s][hf], b3h[s][hf], b3l[s][hf]);
      }
    }
    if (g + 1 < g_end) load_h(hn, g + 1);        // in flight while this group's tiles are worked
    const size_t xoff = (size_t)(px0 / a.x_tile) * a.n_cm * a.x_tile + (px0 % a.x_tile) + 16 * q;
    MfRow<XT> xr[2];
    xr[0].load(x_cm + xoff + (size_t)min(cbase + l16, a.n_cm - 1) * a.x_tile);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      if (t + 1 < CT) xr[(t + 1) & 1].load(x_cm + xoff + (size_t)min(cbase + 16 * (t + 1) + l16, a.n_cm - 1) * a.x_tile);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float x[4], r[4];
        xr[t & 1].quad(s, x);
        if constexpr (L2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = x[i];
        } else {
          mf_f4 y = mf_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int hf = KH - 1; hf >= 0; --hf) y = mf_mma3<L2 ? 0 : 4>(a1h[s][hf], a1l[s][hf], gh[t][hf], gl[t][hf], y);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // The first reader of a matrix-core result must be an instruction the compiler knows: it owes the wait states
            // between v_mfma and a vector read of its result, cannot see into the asm below and places none for it (found in
            // mu_h_mfma_kernel.hpp, where the same pattern gave non-finite numerators with two waves per SIMD).
            float yi = fmaxf(y[i], 1e-37f);
            asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0" : "+v"(yi));   // in place: the transcendental unit reads its source late (the matrix-core hazard); s_nop: the wait state a vector instruction that reads a transcendental result needs - the compiler's hazard recognizer does not see into the asm
            r[i] = x[i] * yi;
          }
        }
        mf_s4 rh, rl;
        mf_split(r, rh, rl);
#pragma unroll
        for (int hf = 0; hf < KH; ++hf) acc[t][hf] = mf_mma3<L2 ? 0 : 8>(rh, rl, b3h[s][hf], b3l[s][hf], acc[t][hf]);
      }
    }
  }
  // A[channels cbase + 16 t + 4 q + r, component 16 hf + l16] -> slab (k, n_pad): 4 consecutive channels per lane
#pragma unroll
  for (int hf = 0; hf < KH; ++hf) {
    if (16 * hf + l16 < K) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c0 = cbase + 16 * t + 4 * q;
        if (c0 < a.n_pad)   // (n_pad is a multiple of 8: a quad is inside or outside as a whole)
          *reinterpret_cast<float4*>(a.a_slab + ((size_t)blockIdx.x * K + 16 * hf + l16) * a.n_pad + c0) = make_float4(acc[t][hf][0], acc[t][hf][1], acc[t][hf][2], acc[t][hf][3]);
      }
    }
  }
}

template <int K, typename XT, bool L2 = false>
__global__ __launch_bounds__(256) void w_accum_mfma_kernel(const WAccumArgs a) {
  static_assert(!L2, "the Frobenius variant is w_accum_mfma_l2_kernel");
  w_accum_mfma_body<K, XT, false>(a);
}
template <int K>
__global__ __launch_bounds__(256) void w_accum_mfma_l2_kernel(const WAccumArgs a) {
  w_accum_mfma_body<K, float, true>(a);
}

}  // namespace espm
