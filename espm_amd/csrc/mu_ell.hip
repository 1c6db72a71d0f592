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
static int launch_h_ell_k(const HStepArgs& args_in, int nblk, hipStream_t stream) {
  constexpr int UNR = K > 8 ? 2 : ESPM_ELL_UNR_H;   // (a batch holds 2 UNR gathered rows of K floats)
  const size_t red = (size_t)(ESPM_ELL_TILE / 64 + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  // [nsplit][K][tile_px]: K * 512 floats whatever the split; two such sets when the groups of a 512-pixel window are walked in pairs
  size_t part = (size_t)K * ESPM_ELL_TILE * sizeof(float) * ((K <= ESPM_ELL_PAIR_MAX_K && args_in.ell_tp == ESPM_ELL_TILE) ? 2 : 1);
  if (red > part) part = red;
  const size_t tail_scratch = (size_t)((ESPM_ELL_TILE / 64 + 1) * (KP + 1) + 1) * sizeof(double);
  if (tail_scratch > part) part = tail_scratch;
  HStepArgs args = args_in;
  size_t bytes = (size_t)args.n_pad * EllTab<K>::FLOATS * sizeof(float) + part;
  if (args.cs_parts) {   // the workgroup's own copy of colsum(GW'): k doubles behind the numerators
    if (bytes + KP * sizeof(double) <= ESPM_ELL_LDS_MAX) {
      args.cs_lds_off = (int)bytes;
      bytes += KP * sizeof(double);
    } else {
      // ... or, where table and numerators take the workgroup's LDS to the last byte (16 components at 2048 channels: 128 + 32 KB), in the
      // table's first bytes once the walk is over and the table dead (the kernel: cs_late)
      args.cs_lds_off = 0;
    }
  }
  const dim3 grid(nblk + (args.tail_on ? 1 : 0));   // (+ the tail of the previous W update, espm_mu_iterate)
  if (args.h_rule == 1 || args.h_rule == 2) {  // quadratic surrogate of the Laplacian term / projected gradient
    auto go = [&](auto kern) -> int {
      if (int rc = allow_lds(kern, bytes, "h_step (ell)")) return rc;
      hipLaunchKernelGGL(kern, grid, dim3(ESPM_ELL_TILE), bytes, stream, args);
      return ESPM_OK;
    };
    int rc;
    if (args.h_rule == 1) rc = args.compute_loss ? go(h_step_ell_kernel<K, true, UNR, 1>) : go(h_step_ell_kernel<K, false, UNR, 1>);
    else rc = args.compute_loss ? go(h_step_ell_kernel<K, true, UNR, 2>) : go(h_step_ell_kernel<K, false, UNR, 2>);
    if (rc) return rc;
  } else if (args.compute_loss) {
    if (int rc = allow_lds(h_step_ell_kernel<K, true, UNR>, bytes, "h_step (ell)")) return rc;
    hipLaunchKernelGGL((h_step_ell_kernel<K, true, UNR>), grid, dim3(ESPM_ELL_TILE), bytes, stream, args);
  } else {
    if (int rc = allow_lds(h_step_ell_kernel<K, false, UNR>, bytes, "h_step (ell)")) return rc;
    hipLaunchKernelGGL((h_step_ell_kernel<K, false, UNR>), grid, dim3(ESPM_ELL_TILE), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "h_step (ell) launch");
}

int launch_h_ell(const HStepArgs& args, int nblk, hipStream_t stream) {
  ESPM_REQUIRE(args.ell && args.ell_off && args.ell_klc && args.ell_pix, "h_step: the sparse store needs ell_h, ell_h_off, ell_klc, pix_perm");
  ESPM_REQUIRE(args.ell_tp == 64 || args.ell_tp == 128 || args.ell_tp == 256 || args.ell_tp == 512, "h_step: sparse store tile_px=%d must be 64, 128, 256 or 512", args.ell_tp);
  ESPM_REQUIRE(args.ell_bits >= 1 && args.ell_bits <= 14 && (1 << args.ell_bits) >= args.n, "h_step: ell_cbits=%d does not cover n=%d", args.ell_bits, args.n);
  switch (args.k) {
#if ESPM_KP <= 16   // (the widest build - 17..32 components - has the dense stores only: a table row of 32 floats leaves the LDS no room)
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
  constexpr int UNR = K > 8 ? 2 : ESPM_ELL_UNR_W;
  // one workgroup per pixel block while the blocks alone cover the chip; otherwise the channel groups of a
  // block are dealt to `csplit` workgroups, each with as many waves (<= 16) as it has groups (two per wave
  // when csplit = 1: heavy / light pairing)
  int csplit = 1;
  while (nblk * csplit < device_cus() && csplit * 2 <= args.n_cg) csplit *= 2;
  const int mine = (args.n_cg + csplit - 1) / csplit;
  int nw = csplit == 1 ? (mine + 1) / 2 : mine;
  if (nw > ESPM_ELL_WTHREADS / 64) nw = ESPM_ELL_WTHREADS / 64;
  if (nw < 1) nw = 1;
  const size_t bytes = (size_t)args.pb * EllTab<K>::FLOATS * sizeof(float);
  if (args.pb == ESPM_ELL_PB) {
    if (int rc = allow_lds(w_accum_ell_kernel<K, UNR, true>, bytes, "w_accum (ell)")) return rc;
    hipLaunchKernelGGL((w_accum_ell_kernel<K, UNR, true>), dim3(nblk, csplit), dim3(64 * nw), bytes, stream, args);
  } else {
    hipLaunchKernelGGL((w_accum_ell_kernel<K, UNR, false>), dim3(nblk, csplit), dim3(64 * nw), bytes, stream, args);
  }
  return check_hip(hipGetLastError(), "w_accum (ell) launch");
}

int launch_w_ell(const WAccumArgs& args, int k, int nblk, hipStream_t stream) {
  ESPM_REQUIRE(args.ell && args.ell_off && args.chan_perm && args.n_cg >= 1, "w_accum: the sparse store needs ell_w, ell_w_off, chan_perm");
  ESPM_REQUIRE(args.pb >= 128 && args.pb <= ESPM_ELL_PB && (args.pb & (args.pb - 1)) == 0 && nblk == (args.p + args.pb - 1) / args.pb,
               "w_accum: nblk_w=%d must be ceil(p / %d) for the sparse store", nblk, args.pb);
  switch (k) {
#if ESPM_KP <= 16
#define ESPM_X(KK) case KK: return launch_w_ell_k<KK>(args, nblk, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
  }
  return set_error(ESPM_EUNSUPPORTED, "w_accum: k=%d not built", k);
}

// Pixels without counts (include/espm_mu.h, ell_fill_*): numerator of the reference's log_shift fill,
// fill * sum_c GW_c / (GW_c . H_pixel) over the n real channels (updates.py:127-132 restricted to that pixel column); one wave
// per listed pixel, lanes over the channels, rows of gw_s straight from L2 (n x KP floats).
__global__ __launch_bounds__(256) void ell_fill_num_kernel(const float* __restrict__ gw_s, const float* __restrict__ h_in,
                                                           const int32_t* __restrict__ fill_px, int fill_n, int n, int k, int p_pad,
                                                           float fill, float* __restrict__ fill_num) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= fill_n) return;   // (whole waves)
  const int px = fill_px[idx];
  float h[KP], s[KP];
#pragma unroll
  for (int kk = 0; kk < KP; ++kk) {
    h[kk] = kk < k ? h_in[(size_t)kk * p_pad + px] : 0.f;
    s[kk] = 0.f;
  }
  for (int c = lane; c < n; c += 64) {
    float g[KP], y = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      g[kk] = gw_s[(size_t)c * KP + kk];
      if (kk < k) y = fmaf(g[kk], h[kk], y);
    }
    const float r = 1.f / y;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) s[kk] = fmaf(g[kk], r, s[kk]);
  }
#pragma unroll
  for (int kk = 0; kk < KP; ++kk) {
    const float t = wave_sum(s[kk]);
    if (lane == 0 && kk < k) fill_num[(size_t)kk * fill_n + idx] = fill * t;
  }
}

int launch_ell_fill_num(const float* gw_s, const float* h_in, const int32_t* fill_px, int fill_n, int n, int k, int p_pad, float fill,
                        float* fill_num, hipStream_t stream) {
  hipLaunchKernelGGL(ell_fill_num_kernel, dim3((fill_n + 3) / 4), dim3(256), 0, stream, gw_s, h_in, fill_px, fill_n, n, k, p_pad, fill,
                     fill_num);
  return check_hip(hipGetLastError(), "ell_fill_num launch");
}

}  // namespace espm
