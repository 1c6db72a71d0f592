// Launchers of the sparse count store kernels (mu_ell_kernel.hpp).
#include "mu_ell_kernel.hpp"

#ifndef ESPM_ELL_UNR_H
#define ESPM_ELL_UNR_H 4
#endif
#ifndef ESPM_ELL_UNR_W
#define ESPM_ELL_UNR_W 4
#endif

namespace espm {

// dynamic LDS above 64 KB has to be granted per kernel (once)
template <typename KernelT>
static int allow_lds(KernelT kern, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return ESPM_OK;
  if (bytes > ESPM_ELL_LDS_MAX) return set_error(ESPM_EUNSUPPORTED, "%s: %zu bytes of LDS exceed %d", what, bytes, ESPM_ELL_LDS_MAX);
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), what);
}

template <int K>
static int launch_h_ell_k(const HStepArgs& args, int nblk, hipStream_t stream) {
  constexpr int UNR = ESPM_ELL_UNR_H;
  const size_t red = (size_t)(ESPM_ELL_TILE / 64 + 1) * (ESPM_HP_NSCALAR + 2 * K) * sizeof(double);
  // [nsplit][K][tile_px]: K * 512 floats whatever the split; two such sets when the groups of a 512-pixel window are walked in pairs
  size_t part = (size_t)K * ESPM_ELL_TILE * sizeof(float) * ((K <= ESPM_ELL_PAIR_MAX_K && args.ell_tp == ESPM_ELL_TILE) ? 2 : 1);
  if (red > part) part = red;
  const size_t bytes = (size_t)args.n_pad * EllTab<K>::FLOATS * sizeof(float) + part;
  if (args.h_rule == 1 || args.h_rule == 2) {  // quadratic surrogate of the Laplacian term / projected gradient
    auto go = [&](auto kern) -> int {
      if (int rc = allow_lds(kern, bytes, "h_step (ell)")) return rc;
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(ESPM_ELL_TILE), bytes, stream, args);
      return ESPM_OK;
    };
    int rc;
    if (args.h_rule == 1) rc = args.compute_loss ? go(h_step_ell_kernel<K, true, UNR, 1>) : go(h_step_ell_kernel<K, false, UNR, 1>);
    else rc = args.compute_loss ? go(h_step_ell_kernel<K, true, UNR, 2>) : go(h_step_ell_kernel<K, false, UNR, 2>);
    if (rc) return rc;
  } else if (args.compute_loss) {
    if (int rc = allow_lds(h_step_ell_kernel<K, true, UNR>, bytes, "h_step (ell)")) return rc;
    hipLaunchKernelGGL((h_step_ell_kernel<K, true, UNR>), dim3(nblk), dim3(ESPM_ELL_TILE), bytes, stream, args);
  } else {
    if (int rc = allow_lds(h_step_ell_kernel<K, false, UNR>, bytes, "h_step (ell)")) return rc;
    hipLaunchKernelGGL((h_step_ell_kernel<K, false, UNR>), dim3(nblk), dim3(ESPM_ELL_TILE), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "h_step (ell) launch");
}

int launch_h_ell(const HStepArgs& args, int nblk, hipStream_t stream) {
  ESPM_REQUIRE(args.ell && args.ell_off && args.ell_klc && args.ell_pix, "h_step: the sparse store needs ell_h, ell_h_off, ell_klc, pix_perm");
  ESPM_REQUIRE(args.ell_tp == 64 || args.ell_tp == 128 || args.ell_tp == 256 || args.ell_tp == 512, "h_step: sparse store tile_px=%d must be 64, 128, 256 or 512", args.ell_tp);
  ESPM_REQUIRE(args.ell_bits >= 1 && args.ell_bits <= 14 && (1 << args.ell_bits) >= args.n, "h_step: ell_cbits=%d does not cover n=%d", args.ell_bits, args.n);
  switch (args.k) {
#if ESPM_MIN_K <= 8   // (the LDS table holds rows of at most 8 floats: the wide build has no sparse store)
#define ESPM_X(KK) case KK: return launch_h_ell_k<KK>(args, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
  }
  return set_error(ESPM_EUNSUPPORTED, "h_step: k=%d not built", args.k);
}

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

template <int K>
static int launch_w_ell_k(const WAccumArgs& args, int nblk, hipStream_t stream) {
  constexpr int UNR = ESPM_ELL_UNR_W;
  // one workgroup per pixel block while the blocks alone cover the chip; otherwise the channel groups of a
  // block are dealt to `csplit` workgroups, each with as many waves (<= 16) as it has groups (two per wave
  // when csplit = 1: heavy / light pairing)
  int csplit = 1;
  while (nblk * csplit < device_cus() && csplit * 2 <= args.n_cg) csplit *= 2;
  const int mine = (args.n_cg + csplit - 1) / csplit;
  int nw = csplit == 1 ? (mine + 1) / 2 : mine;
  if (nw > ESPM_ELL_WTHREADS / 64) nw = ESPM_ELL_WTHREADS / 64;
  if (nw < 1) nw = 1;
  const size_t bytes = (size_t)ESPM_ELL_PB * EllTab<K>::FLOATS * sizeof(float);
  hipLaunchKernelGGL((w_accum_ell_kernel<K, UNR>), dim3(nblk, csplit), dim3(64 * nw), bytes, stream, args);
  return check_hip(hipGetLastError(), "w_accum (ell) launch");
}

int launch_w_ell(const WAccumArgs& args, int k, int nblk, hipStream_t stream) {
  ESPM_REQUIRE(args.ell && args.ell_off && args.chan_perm && args.n_cg >= 1, "w_accum: the sparse store needs ell_w, ell_w_off, chan_perm");
  ESPM_REQUIRE(nblk == (args.p + ESPM_ELL_PB - 1) / ESPM_ELL_PB, "w_accum: nblk_w=%d must be ceil(p / %d) for the sparse store", nblk, ESPM_ELL_PB);
  switch (k) {
#if ESPM_MIN_K <= 8
#define ESPM_X(KK) case KK: return launch_w_ell_k<KK>(args, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
  }
  return set_error(ESPM_EUNSUPPORTED, "w_accum: k=%d not built", k);
}

}  // namespace espm
