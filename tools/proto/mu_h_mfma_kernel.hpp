// H-step kernel with the product Y = GW H on the matrix cores (v_mfma_f32_16x16x32_bf16).
//
// The VALU kernel (mu_h_kernel.hpp) spends 5 of its 12 fused multiply-adds per X element on Y.  Here a wave
// owns a 16-channel x 128-pixel block per step: eight 16 x 16 x 32 MFMAs form Y for it, and the vector ALU
// keeps only the reciprocal, the numerator accumulation and the loss term.
//
// fp32-grade operands from bf16 matrix cores: every fp32 value v is split into three bf16 terms
// v = hi + mid + lo (each split exact, |lo| <= 2^-16 |v|), and the six products hi*hi, hi*mid, mid*hi,
// hi*lo, lo*hi, mid*mid of one component occupy six of the 32 k-slots of the instruction:
//     Y[c, j] = sum_{q < 6, kk < K}  GWsplit[pa(q)][c, kk] * Hsplit[pb(q)][kk, j]
// (6 K <= 32 for K <= 5: one MFMA per 16 x 16 tile; K <= 8 uses two).  The dropped terms are below 2^-23
// relative; the MFMA accumulates in fp32.
//
// Register layout (C/D map of the instruction: col = lane & 15, row = 4 (lane >> 4) + reg): lane (g, j) gets
// Y[4g + i][column j] for i < 4.  Sub-tile t < 8 maps its column j to pixel 8 j + t, so after the eight MFMAs
// the lane holds 4 channels x 8 CONSECUTIVE pixels - exactly what four coalesced row loads of X deliver
// (8 pixels per lane and channel: one 8-byte load for 8-bit counts, 16 bytes for bf16).
// Pairs for v_pk_fma_f32 are pixel pairs (2u, 2u+1) of one channel: adjacent elements of the X row; the two Y
// entries come from two accumulator tiles (the accumulator reads place them in one register pair).
#pragma once
// (retired from the product in round 1; kept as a record - it needs the product headers of that
//  time plus the helpers below, which moved here from mu_common.hpp when the variant was retired.  Round 2's matrix-core
//  H-step for 13..16 components is espm_amd/csrc/mu_h_mfma_kernel.hpp; the instability this variant was retired for was
//  very likely the one found and fixed there: an inline-asm reciprocal as the FIRST reader of a v_mfma result gets no
//  wait states from the compiler.)
#include <type_traits>

#include "mu_h_kernel.hpp"

namespace espm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (split3 / build_slots / MfmaCount: mu_common.hpp, shared with the W finish that builds the A fragments)
// XT: uint8_t or bf16_t storage (8 pixels per lane and channel row), NW waves split the channel blocks.
template <int K, typename XT, int NW, bool LOSS>
__global__ __launch_bounds__(NW * 64) void h_step_mfma_kernel(const HStepArgs a) {
  constexpr int NMF = MfmaCount<K>::value;
  constexpr int TP = 128;                 // pixels per workgroup tile
  constexpr int SLOTS = 32 * NMF;
  constexpr int BROW = SLOTS + 8;         // padded LDS row (bf16 elements) of the H-split image
  extern __shared__ __attribute__((aligned(16))) float smem[];  // H splits, later [NW * 4][K][TP] partial numerators
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int tile0 = blockIdx.x * TP;

  // ---- B operands: the bf16 split image of H for this pixel tile, through LDS -----------------------------
  uint16_t* bimg = reinterpret_cast<uint16_t*>(smem);
  for (int px = threadIdx.x; px < TP; px += NW * 64) {
    float hv[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) hv[kk] = a.h_in[(size_t)kk * a.p_pad + tile0 + px];
    uint16_t slots[SLOTS];
    build_slots<K, NMF>(hv, 1, slots);
#pragma unroll
    for (int s = 0; s < SLOTS; s += 8) {
      uint4 v;
      v.x = slots[s] | ((uint32_t)slots[s + 1] << 16);
      v.y = slots[s + 2] | ((uint32_t)slots[s + 3] << 16);
      v.z = slots[s + 4] | ((uint32_t)slots[s + 5] << 16);
      v.w = slots[s + 6] | ((uint32_t)slots[s + 7] << 16);
      // row (px % 8) * 16 + px / 8: the 16 lanes j of one fragment read consecutive rows (80-byte stride:
      // conflict-free ds_read_b128), not rows 8 apart
      *reinterpret_cast<uint4*>(bimg + (size_t)((px & 7) * 16 + (px >> 3)) * BROW + s) = v;
    }
  }
  __syncthreads();
  // the image stays in LDS for the whole channel loop: every block re-reads its 8 fragments (16 B per lane and
  // sub-tile) instead of pinning 32 registers per lane - occupancy matters more than LDS bandwidth here
  const int brow_off = j * BROW + 8 * g;  // sub-tile t: pixel 8 j + t = row 16 t + j, k-slots 8 g .. 8 g + 7 (+ 32 m)

  f2 acc[K][4];     // [component][pixel pair u]: pixels (2u, 2u + 1), summed over this lane's channels
  f2 kl = {0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < K; ++kk)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[kk][u] = f2{0.f, 0.f};

  const int nblocks = a.n_cm / 16;
  const XT* xtile = static_cast<const XT*>(a.x_cm) + (size_t)(tile0 / a.x_tile) * a.n_cm * a.x_tile + (tile0 % a.x_tile) + 8 * j;
  const uint4* ga = reinterpret_cast<const uint4*>(a.gw_a);

  // operands of one 16-channel block
  struct Blk {
    bf16x8 af[NMF];
    float gflat[4 * K];   // fp32 GW of my four channels, stored as two channel pairs: gwp[(c / 2)][kk][2]
    XVec<XT, 8> xr[4];
  };
  auto load_blk = [&](Blk& b, int blk) {
#pragma unroll
    for (int m = 0; m < NMF; ++m) b.af[m] = __builtin_bit_cast(bf16x8, ga[((size_t)blk * NMF + m) * 64 + lane]);
    const float4* gp4 = reinterpret_cast<const float4*>(a.gw_p + (size_t)(blk * 8 + 2 * g) * 2 * K);
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const float4 v = gp4[q];
      b.gflat[4 * q] = v.x; b.gflat[4 * q + 1] = v.y; b.gflat[4 * q + 2] = v.z; b.gflat[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) b.xr[i].load(xtile + (size_t)(blk * 16 + 4 * g + i) * a.x_tile);
  };
  auto compute_blk = [&](const Blk& b) {
    // opaque copy of the (loop-invariant) LDS address: keeps the fragment reads inside the loop, otherwise
    // the compiler hoists all of them into 32 registers per lane and the kernel drops to 2 waves per SIMD
    int boff = brow_off;
    asm volatile("" : "+v"(boff));  // (an offset, not the pointer: the reads must stay ds_read, not flat)
    const uint16_t* bp = bimg + boff;
    f32x4 y[8];  // Y[channel 4g + i][pixel 8j + t] = y[t][i]
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NMF; ++m) y[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            b.af[m], __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bp + (size_t)(16 * t) * BROW + 32 * m)), y[t], 0, 0, 0);
    }
    // pixel pairs (2u, 2u + 1) of one channel share an fp32x2 register: adjacent elements of the X row, the
    // matching Y entries are gathered from two accumulator tiles by the accumulator reads themselves
    auto channel = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float gv[K];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) gv[kk] = b.gflat[(i >> 1) * 2 * K + 2 * kk + (i & 1)];
      auto pixel_pair = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const f2 yy = f2{y[2 * u][i], y[2 * u + 1][i]};
        const f2 xx = f2{b.xr[i].template elem<2 * u>(), b.xr[i].template elem<2 * u + 1>()};
        // In-place reciprocals (source = destination register).  v_accvgpr_read_b32 is NOT ordered against the
        // operand read of an earlier transcendental op: with two waves on a SIMD the compiler's
        //   v_rcp_f32 v72, v55 ... v_accvgpr_read_b32 v55, a8
        // let the accumulator read overwrite v55 before the rcp had read lanes 48-63 of it (measured: wrong
        // numerators for g = 3, even pixels only, non-deterministic; DESIGN.md "matrix-core variant").  With
        // source = destination the next writer is ordered behind the rcp's own result write.
        f2 inv = yy;
        asm("v_rcp_f32_e32 %0, %0\n\tv_rcp_f32_e32 %1, %1\n\ts_nop 0" : "+v"(inv.x), "+v"(inv.y));
        const f2 r = LOSS ? xx * inv + f2{1e-37f, 1e-37f} : xx * inv;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) acc[kk][u] = gv[kk] * r + acc[kk][u];
        if constexpr (LOSS) {
          f2 lg = r;  // r is dead after the accumulation above: log in place, same ordering argument as for rcp
          asm("v_log_f32_e32 %0, %0\n\tv_log_f32_e32 %1, %1\n\ts_nop 0" : "+v"(lg.x), "+v"(lg.y));
          kl = xx * lg + kl;
        }
      };
      pixel_pair(std::integral_constant<int, 0>{});
      pixel_pair(std::integral_constant<int, 1>{});
      pixel_pair(std::integral_constant<int, 2>{});
      pixel_pair(std::integral_constant<int, 3>{});
    };
    channel(std::integral_constant<int, 0>{});
    channel(std::integral_constant<int, 1>{});
    channel(std::integral_constant<int, 2>{});
    channel(std::integral_constant<int, 3>{});
  };

  // (explicit double buffering of the block operands was measured: it costs the third wave per SIMD through
  //  register pressure and ran slower - 297 vs 208 us at the headline size)
  for (int blk = wave; blk < nblocks; blk += NW) {
    Blk b;
    load_blk(b, blk);
    compute_blk(b);
  }

  __syncthreads();  // every wave is done with the H-split image: smem is reused for the partial numerators
  // ---- partial numerators of (wave, g) to LDS: [NW * 4][K][TP] ------------------------------------------------
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    float* dst = smem + ((size_t)(wave * 4 + g) * K + kk) * TP + 8 * j;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dst[2 * u] = acc[kk][u].x;
      dst[2 * u + 1] = acc[kk][u].y;
    }
  }
  h_epilogue<K, false>(a, smem, NW * 4, TP, tile0, LOSS ? kl.x + kl.y : 0.f);
}

}  // namespace espm


// ---- helpers that lived in mu_common.hpp while the variant was part of the library ----------------------------
#if 0
// ---- 3-way bf16 splits for the matrix-core product Y = GW H (mu_h_mfma_kernel.hpp) ----------------------
// which bf16 term of GW (A side) and of H (B side) product group q uses: 0 = hi, 1 = mid, 2 = lo
__host__ __device__ constexpr int split_a(int q) { return q == 2 ? 1 : (q == 4 ? 2 : (q == 5 ? 1 : 0)); }
__host__ __device__ constexpr int split_b(int q) { return q == 1 ? 1 : (q == 3 ? 2 : (q == 5 ? 1 : 0)); }

__device__ __forceinline__ uint16_t bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
// v = t[0] + t[1] + t[2] with bf16 terms (the residuals are exact in fp32)
__device__ __forceinline__ void split3(float v, uint16_t (&t)[3]) {
  t[0] = bf16_rne(v);
  const float r1 = v - bf16_f32(t[0]);
  t[1] = bf16_rne(r1);
  t[2] = bf16_rne(r1 - bf16_f32(t[1]));
}
// the 32 * NMF k-slots of one row of GW (side = 0) or one column of H (side = 1); slot q * K + kk
template <int K, int NMF>
__device__ __forceinline__ void build_slots(const float (&v)[K], int side, uint16_t (&slots)[32 * NMF]) {
  uint16_t parts[K][3];
#pragma unroll
  for (int kk = 0; kk < K; ++kk) split3(v[kk], parts[kk]);
#pragma unroll
  for (int s = 0; s < 32 * NMF; ++s) slots[s] = 0;
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) slots[q * K + kk] = side == 0 ? parts[kk][split_a(q)] : parts[kk][split_b(q)];
}

template <int K>
struct MfmaCount {
  static constexpr int value = (6 * K + 31) / 32;
};


#endif
