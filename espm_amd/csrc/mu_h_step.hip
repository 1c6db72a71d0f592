// H update of the SmoothNMF multiplicative rule, fused over one pixel tile per workgroup.
//
// Restates espm/estimators/updates.py:83-156 (KL branch) without materialising any (n, p)
// intermediate:  Y = GW H,  R = X / Y,  num = GW^T R  are formed per pixel column in registers,
// with lanes <-> pixels (so the contraction over channels needs no cross-lane traffic) and the
// channel range split over the waves of the workgroup.  X is streamed once, channel-major, with
// 16-byte (bf16 x 8) coalesced loads; GW rows are wave-uniform and come through the scalar cache.
// The epilogue (one thread per pixel) adds the log-sparsity and Laplacian terms, solves the
// per-pixel simplex multiplier (dicotomy.py:4-55) in registers, clamps and writes H', its
// transposed copy for the W-step, and per-workgroup partial sums:
//   - KL data term of the INPUT state: sum X log2(X / Y)   (measures.py:493-504, base.py:200-203)
//   - log regulariser and Laplacian quadratic form of the input state (measures.py:543-548, :574-577)
//   - row sums / row maxima of H' (updates.py:60, :139 of the NEXT half steps)
#include "mu_h_kernel.hpp"
#include "mu_h_mfma_kernel.hpp"

// both contractions of the dense H-step on the matrix cores from this many components on (the wide build)
#ifndef ESPM_H_MFMA_MIN_K
#define ESPM_H_MFMA_MIN_K 13   // measured at the headline image, 8-bit store (profiles/r02o_wide_h_mfma_ab.log): k = 12 665 vs 651 us per iteration, k = 13 671 vs 686, k = 16 687 vs 759
#endif

// This file holds most of the library's kernel instantiations (component count x store x tile x rule); the build may
// compile it ESPM_H_PARTS (<= 4) times, part ESPM_H_PART instantiating every ESPM_H_PARTS-th component count, so that
// the parts compile side by side.  Part 0 carries everything that is not a template.
#ifndef ESPM_H_PARTS
#define ESPM_H_PARTS 1
#define ESPM_H_PART 0
#endif
#define ESPM_CAT2(a, b) a##b
#define ESPM_CAT(a, b) ESPM_CAT2(a, b)

// From this many components on only the instances WITH the loss terms are built (the widest build: 16 component counts of kernels that
// unroll over K - compile time); a launch that did not ask for the loss gets it computed all the same (the `false` instance below IS the `true` one there), the state and the records are the same.
#define ESPM_H_LOSS_ALWAYS_ABOVE 16

namespace espm {

#if ESPM_H_PART == 0
// ---- reduction of the per-workgroup records (one workgroup per value: h_finalize_one) --------
__global__ __launch_bounds__(256) void h_finalize_kernel(const HFinalizeArgs a) {
  __shared__ double scratch[5 * (ESPM_HP_NSCALAR + 2 * KP + 1)];
  h_finalize_one(a, (int)blockIdx.x, scratch);
}
#endif

// ---- dispatch ---------------------------------------------------------------------------------
// Partial numerators of more than 16 components (the widest build) pass the 64 KB of LDS a kernel gets without asking: granted per
// kernel, up to the CU's 160 KB (K * 4096 bytes with the vector kernels' tiles: 128 KB at 32 components).
template <typename KernelT>
static int allow_lds(KernelT kern, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return ESPM_OK;
  if (bytes > 160 * 1024) return set_error(ESPM_EUNSUPPORTED, "%s: %zu bytes of LDS exceed a CU's 160 KB", what, bytes);
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), what);
}

template <int K, typename XT, int PX, int NW, int U, int NBUF>
static int launch_h(const HStepArgs& args, int nblk, hipStream_t stream) {
  const size_t lds = (size_t)NW * K * 64 * PX * sizeof(float);
  const size_t lds_min = (size_t)(NW + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  const size_t bytes = lds > lds_min ? lds : lds_min;
  if (args.compute_loss) {
    if (int rc = allow_lds(h_step_kernel<K, XT, PX, NW, true, U, NBUF>, bytes, "h_step")) return rc;
    hipLaunchKernelGGL((h_step_kernel<K, XT, PX, NW, true, U, NBUF>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  } else {
    if (int rc = allow_lds(h_step_kernel<K, XT, PX, NW, (K > ESPM_H_LOSS_ALWAYS_ABOVE), U, NBUF>, bytes, "h_step")) return rc;
    hipLaunchKernelGGL((h_step_kernel<K, XT, PX, NW, (K > ESPM_H_LOSS_ALWAYS_ABOVE), U, NBUF>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "h_step launch");
}

template <int K, typename XT, int TILE>
static int launch_h_mfma(const HStepArgs& args, int nblk, hipStream_t stream) {
  constexpr int STEPS = ESPM_H_MFMA_STEPS, PASSES = TILE / (16 * STEPS);   // pixel steps of 16 per pass, passes per tile
  const size_t lds = (size_t)4 * K * 16 * STEPS * PASSES * sizeof(float);
  const size_t lds_min = (size_t)(4 + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  size_t bytes = lds > lds_min ? lds : lds_min;
#ifdef ESPM_H_MFMA_PAD_LDS   // diagnosis (mu_w_mfma_kernel.hpp, the 32-slot form): the SAME code with one workgroup per CU, by an LDS request nobody uses
  bytes += ESPM_H_MFMA_PAD_LDS;
  if (bytes > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h_step_mfma_kernel<K, XT, STEPS, PASSES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h_step_mfma_kernel<K, XT, STEPS, PASSES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  }
#endif
  if (args.compute_loss) {
    if (int rc = allow_lds(h_step_mfma_kernel<K, XT, STEPS, PASSES, true>, bytes, "h_step (mfma)")) return rc;
    hipLaunchKernelGGL((h_step_mfma_kernel<K, XT, STEPS, PASSES, true>), dim3(nblk), dim3(256), bytes, stream, args);
  } else {
    if (int rc = allow_lds(h_step_mfma_kernel<K, XT, STEPS, PASSES, (K > ESPM_H_LOSS_ALWAYS_ABOVE)>, bytes, "h_step (mfma)")) return rc;
    hipLaunchKernelGGL((h_step_mfma_kernel<K, XT, STEPS, PASSES, (K > ESPM_H_LOSS_ALWAYS_ABOVE)>), dim3(nblk), dim3(256), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "h_step (mfma) launch");
}

template <int K, int PX, int NW, int U, int NBUF>
static int launch_h_l2(const HStepArgs& args, int nblk, hipStream_t stream) {
  const size_t lds = (size_t)NW * K * 64 * PX * sizeof(float);
  const size_t lds_min = (size_t)(NW + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  if (int rc = allow_lds(h_step_kernel<K, float, PX, NW, false, U, NBUF, true>, lds > lds_min ? lds : lds_min, "h_step (l2)")) return rc;
  hipLaunchKernelGGL((h_step_kernel<K, float, PX, NW, false, U, NBUF, true>), dim3(nblk), dim3(NW * 64), lds > lds_min ? lds : lds_min,
                     stream, args);
  return check_hip(hipGetLastError(), "h_step (l2) launch");
}

template <int K, int PX, int NW, int U, int NBUF, int RULE>
static int launch_h_rule(const HStepArgs& args, int nblk, hipStream_t stream) {
  const size_t lds = (size_t)NW * K * 64 * PX * sizeof(float);
  const size_t lds_min = (size_t)(NW + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  const size_t bytes = lds > lds_min ? lds : lds_min;
  if (args.compute_loss) {
    if (int rc = allow_lds(h_step_kernel<K, float, PX, NW, true, U, NBUF, false, RULE>, bytes, "h_step (alternate rule)")) return rc;
    hipLaunchKernelGGL((h_step_kernel<K, float, PX, NW, true, U, NBUF, false, RULE>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  } else {
    if (int rc = allow_lds(h_step_kernel<K, float, PX, NW, (K > ESPM_H_LOSS_ALWAYS_ABOVE), U, NBUF, false, RULE>, bytes, "h_step (alternate rule)")) return rc;
    hipLaunchKernelGGL((h_step_kernel<K, float, PX, NW, (K > ESPM_H_LOSS_ALWAYS_ABOVE), U, NBUF, false, RULE>), dim3(nblk), dim3(NW * 64), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "h_step (alternate rule) launch");
}

template <int K>
static int dispatch_h_k(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  if (args.h_rule != 0) {  // quadratic surrogate / projected gradient: dense stores through the fp32 one
    if (x_dtype != ESPM_X_F32) return set_error(ESPM_EUNSUPPORTED, "h_rule %d is built for the sparse and the f32 store", args.h_rule);
    if constexpr (K <= 16) if (args.h_rule == 1 && tile_px == 256) return launch_h_rule<K, 4, 4, 8, 0, 1>(args, nblk, stream);
    if (args.h_rule == 1 && tile_px == 128) return launch_h_rule<K, 2, 8, 8, 2, 1>(args, nblk, stream);
    if constexpr (K <= 16) if (args.h_rule == 2 && tile_px == 256) return launch_h_rule<K, 4, 4, 8, 0, 2>(args, nblk, stream);
    if (args.h_rule == 2 && tile_px == 128) return launch_h_rule<K, 2, 8, 8, 2, 2>(args, nblk, stream);
    return set_error(ESPM_EINVAL, "h_step (rule %d): tile_px %d not available", args.h_rule, tile_px);
  }
  if (args.l2_m) {  // Frobenius branch: fp32 store only
    if (x_dtype != ESPM_X_F32 || args.compute_loss) return set_error(ESPM_EUNSUPPORTED, "the l2 H-step needs the f32 store and no loss");
    if constexpr (K <= 16) if (tile_px == 256) return launch_h_l2<K, 4, 4, 8, 0>(args, nblk, stream);
    if (tile_px == 128) return launch_h_l2<K, 2, 8, 8, 2>(args, nblk, stream);
    return set_error(ESPM_EINVAL, "h_step (l2): tile_px %d not available", tile_px);
  }
  if constexpr (K >= ESPM_H_MFMA_MIN_K) {   // matrix cores (mu_h_mfma_kernel.hpp): the pixel-major copy of X, tiles of 16 pixel steps
    if constexpr (K <= 16) {
      if (args.mfma && args.x_pm && tile_px == 256) {
        if (x_dtype == ESPM_X_U8) return launch_h_mfma<K, uint8_t, 256>(args, nblk, stream);
        if (x_dtype == ESPM_X_BF16) return launch_h_mfma<K, bf16_t, 256>(args, nblk, stream);
        return launch_h_mfma<K, float, 256>(args, nblk, stream);
      }
    }
    if (args.mfma && args.x_pm && tile_px == 128) {   // (the widest build: 128-pixel tiles only - espm_mu_query; 4 waves' partial numerators of 32 components and 128 pixels are the 64 KB of LDS a kernel gets without asking)
      if (x_dtype == ESPM_X_U8) return launch_h_mfma<K, uint8_t, 128>(args, nblk, stream);
      if (x_dtype == ESPM_X_BF16) return launch_h_mfma<K, bf16_t, 128>(args, nblk, stream);
      return launch_h_mfma<K, float, 128>(args, nblk, stream);
    }
  }
  if constexpr (K > 16) {   // the widest build keeps the vector kernels for the fp32 store only (the alternate rules, the Frobenius branch, A/B)
    if (x_dtype != ESPM_X_F32)
      return set_error(ESPM_EUNSUPPORTED, "h_step: %d components on the 8-bit / bf16 store run on the matrix cores only (pixel-major copy of X, no_fused = 0, tile_px 128)", K);
  }
  if (x_dtype == ESPM_X_U8) {
    if constexpr (K <= 16) {
      if (tile_px == 256) return launch_h<K, uint8_t, 4, 4, 8, 0>(args, nblk, stream);
      if (tile_px == 128) return launch_h<K, uint8_t, 2, 8, 8, 2>(args, nblk, stream);
    }
  } else if (x_dtype == ESPM_X_BF16) {
    if constexpr (K <= 16) {
      if (tile_px == 256) return launch_h<K, bf16_t, 4, 4, 8, 0>(args, nblk, stream);
      if (tile_px == 128) return launch_h<K, bf16_t, 2, 8, 8, 2>(args, nblk, stream);
    }
  } else {
    if constexpr (K <= 16) if (tile_px == 256) return launch_h<K, float, 4, 4, 8, 0>(args, nblk, stream);
    if (tile_px == 128) return launch_h<K, float, 2, 8, 8, 2>(args, nblk, stream);
  }
  return set_error(ESPM_EINVAL, "h_step: tile_px %d not available for x_dtype %d", tile_px, x_dtype);
}

template <int KK>
static int dispatch_h_part_k(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  if constexpr ((KK - ESPM_MIN_K) % ESPM_H_PARTS == ESPM_H_PART) return dispatch_h_k<KK>(args, x_dtype, tile_px, nblk, stream);
  else return set_error(ESPM_EUNSUPPORTED, "h_step: k=%d belongs to another part of the build", KK);
}

int ESPM_CAT(dispatch_h_step_part, ESPM_H_PART)(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  switch (args.k) {
#define ESPM_X(KK) case KK: return dispatch_h_part_k<KK>(args, x_dtype, tile_px, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
  }
  return set_error(ESPM_EUNSUPPORTED, "h_step: k=%d not built (%d..%d)", args.k, ESPM_MIN_K, ESPM_MAX_K);
}

#if ESPM_H_PART == 0
int dispatch_h_step_part1(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream);
int dispatch_h_step_part2(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream);
int dispatch_h_step_part3(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream);

int dispatch_h_step(const HStepArgs& args, int x_dtype, int tile_px, int nblk, hipStream_t stream) {
  static_assert(ESPM_H_PARTS >= 1 && ESPM_H_PARTS <= 4, "ESPM_H_PARTS: 1..4");
  if (args.k < ESPM_MIN_K || args.k > ESPM_MAX_K)
    return set_error(ESPM_EUNSUPPORTED, "h_step: k=%d not built (%d..%d)", args.k, ESPM_MIN_K, ESPM_MAX_K);
  switch ((args.k - ESPM_MIN_K) % ESPM_H_PARTS) {
#if ESPM_H_PARTS > 1
    case 1: return dispatch_h_step_part1(args, x_dtype, tile_px, nblk, stream);
#endif
#if ESPM_H_PARTS > 2
    case 2: return dispatch_h_step_part2(args, x_dtype, tile_px, nblk, stream);
#endif
#if ESPM_H_PARTS > 3
    case 3: return dispatch_h_step_part3(args, x_dtype, tile_px, nblk, stream);
#endif
  }
  return dispatch_h_step_part0(args, x_dtype, tile_px, nblk, stream);
}

int launch_h_finalize(const HFinalizeArgs& args, hipStream_t stream) {
  hipLaunchKernelGGL(h_finalize_kernel, dim3(H_FINALIZE_JOBS), dim3(256), 0, stream, args);
  return check_hip(hipGetLastError(), "h_finalize launch");
}
#endif

}  // namespace espm
