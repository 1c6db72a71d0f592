// The fused half-steps' common case as instances of their own (mu_fused_kernel.hpp, template parameter PLAIN): simplex over H, Laplacian
// on an image grid, a previous H to compare with, no fixed_H / fill numerators / Bregman / Frobenius, the staged prologue, dynamic
// units, the slab collected in LDS.  launch_fused_k (mu_fused.hip) checks those facts on the host and calls launch_fused_plain; anything
// else takes the generic instance.  A translation unit of its own so that the two sets of instances compile side by side.
// ESPM_PLAIN_STREAM = 1 (mu_fused_stream.hip includes this file so): the full geometry's instances with streamed lists
// (mu_ell_kernel.hpp: ell_list_load), behind launch_fused_plain_stream - a third unit compiling beside the other two.
#include "mu_fused_kernel.hpp"

#ifndef ESPM_PLAIN_STREAM
#define ESPM_PLAIN_STREAM 0
#endif
#if ESPM_PLAIN_STREAM
#define ESPM_PLAIN_ENTRY launch_fused_plain_stream
#define ESPM_PLAIN_PHASE phase_buffer_plain_stream
#else
#define ESPM_PLAIN_ENTRY launch_fused_plain
#define ESPM_PLAIN_PHASE phase_buffer_plain
#endif

#ifndef ESPM_ELL_UNR_H
#define ESPM_ELL_UNR_H 4
#endif
#ifndef ESPM_ELL_UNR_W
#define ESPM_ELL_UNR_W 4
#endif
#ifndef ESPM_FUSED_SMALL_UNR_H
#define ESPM_FUSED_SMALL_UNR_H ESPM_ELL_UNR_H
#endif
#ifndef ESPM_FUSED_SMALL_UNR_W
#define ESPM_FUSED_SMALL_UNR_W ESPM_ELL_UNR_W
#endif

namespace espm {

#if ESPM_MIN_K <= 8
template <int K>
static int launch_plain_k(const FusedArgs& args, bool loss, bool full, int nblk, size_t bytes, hipStream_t stream) {
  auto go = [&](auto kern) -> int {
    if (bytes > 64 * 1024)
      if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                             "fused half-steps (plain)"))
        return rc;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(ESPM_ELL_WTHREADS), bytes, stream, args);
    return check_hip(hipGetLastError(), "fused half-steps (plain) launch");
  };
  if (full)
    return loss ? go(mu_fused_ell_kernel<K, true, ESPM_ELL_UNR_H, ESPM_ELL_UNR_W, ESPM_ELL_WTHREADS, true, true, ESPM_PLAIN_STREAM != 0>)
                : go(mu_fused_ell_kernel<K, false, ESPM_ELL_UNR_H, ESPM_ELL_UNR_W, ESPM_ELL_WTHREADS, true, true, ESPM_PLAIN_STREAM != 0>);
#if ESPM_PLAIN_STREAM
  return set_error(ESPM_EUNSUPPORTED, "fused half-steps (plain, streamed lists): the full geometry only");
#else
  return loss ? go(mu_fused_ell_kernel<K, true, ESPM_FUSED_SMALL_UNR_H, ESPM_FUSED_SMALL_UNR_W, ESPM_ELL_WTHREADS, false, true>)
              : go(mu_fused_ell_kernel<K, false, ESPM_FUSED_SMALL_UNR_H, ESPM_FUSED_SMALL_UNR_W, ESPM_ELL_WTHREADS, false, true>);
#endif
}
#endif

int ESPM_PLAIN_ENTRY(const FusedArgs& args, int k, bool loss, bool full, int nblk, size_t lds_bytes, hipStream_t stream) {
  switch (k) {
#if ESPM_MIN_K <= 8
#define ESPM_X(KK) case KK: return launch_plain_k<KK>(args, loss, full, nblk, lds_bytes, stream);
    ESPM_K_CASES(ESPM_X)
#undef ESPM_X
#endif
  }
  return set_error(ESPM_EUNSUPPORTED, "fused half-steps (plain): k=%d not built", k);
}

#ifdef ESPM_PHASE_CLOCK
// (debug build: this translation unit has its own copy of the stamp buffer's pointer - device globals are per unit without -fgpu-rdc)
int ESPM_PLAIN_PHASE(unsigned long long* p) { return check_hip(hipMemcpyToSymbol(HIP_SYMBOL(espm_phase_buf), &p, sizeof(p)), "phase buffer (plain)"); }
#endif

}  // namespace espm
