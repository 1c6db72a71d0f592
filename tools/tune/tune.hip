// Tuning harness (NOT part of the product library): instantiates variants of the two streaming kernels
// for K = 5 / bf16 next to the product code and exposes them by index.  Built by tools/tune/run_tune.py.
#include "mu_h_kernel.hpp"
#include "mu_w_kernel.hpp"

using namespace espm;

#ifdef TUNE_U8
typedef uint8_t xt_t;
#else
typedef bf16_t xt_t;
#endif

namespace {

template <int PX, int NW, int U, int NBUF>
int run_h(const HStepArgs& args, int p, hipStream_t stream) {
  constexpr int K = 5;
  const int nblk = (p + 64 * PX - 1) / (64 * PX);
  const size_t lds = (size_t)NW * K * 64 * PX * sizeof(float);
  const size_t lds_min = (size_t)(NW + 1) * (ESPM_HP_NSCALAR + 2 * K + 1) * sizeof(double);
  if (args.compute_loss)
    hipLaunchKernelGGL((h_step_kernel<K, xt_t, PX, NW, true, U, NBUF>), dim3(nblk), dim3(NW * 64), lds > lds_min ? lds : lds_min, stream, args);
  else
    hipLaunchKernelGGL((h_step_kernel<K, xt_t, PX, NW, false, U, NBUF>), dim3(nblk), dim3(NW * 64), lds > lds_min ? lds : lds_min, stream, args);
  return (int)hipGetLastError();
}

template <int UP, int NBUF>
int run_w(const WAccumArgs& args, int nblk, hipStream_t stream) {
  dim3 grid(nblk, (args.n_pad + 2047) / 2048);
  hipLaunchKernelGGL((w_accum_kernel<5, xt_t, 8, UP, NBUF>), grid, dim3(256), 0, stream, args);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

const char* tune_h_name(int v) {
  static const char* names[] = {"px4 nw4 u4 nb2", "px4 nw4 u4 nb3", "px4 nw4 u4 nb4", "px4 nw4 u2 nb4", "px4 nw4 u2 nb6",
                                "px4 nw4 u2 nb8", "px4 nw4 u8 nb2", "px4 nw4 u8 nb3", "px4 nw8 u4 nb3", "px4 nw8 u4 nb4",
                                "px8 nw4 u2 nb3", "px8 nw4 u2 nb4", "px8 nw4 u4 nb3", "px2 nw8 u8 nb3", "px4 nw4 u4 nb6",
                                "px4 nw4 u8 nb0"};
  return (v >= 0 && v < 16) ? names[v] : nullptr;
}

// hpart must hold ceil(p / 128) records; returns the tile size used through *tile_px
int tune_h(const espm_mu_state* st, int src, int write_h, int v, int* tile_px, void* stream) {
  HStepArgs a = make_h_args(st, src, write_h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (v) {
    case 0: *tile_px = 256; return run_h<4, 4, 4, 2>(a, st->p, s);
    case 1: *tile_px = 256; return run_h<4, 4, 4, 3>(a, st->p, s);
    case 2: *tile_px = 256; return run_h<4, 4, 4, 4>(a, st->p, s);
    case 3: *tile_px = 256; return run_h<4, 4, 2, 4>(a, st->p, s);
    case 4: *tile_px = 256; return run_h<4, 4, 2, 6>(a, st->p, s);
    case 5: *tile_px = 256; return run_h<4, 4, 2, 8>(a, st->p, s);
    case 6: *tile_px = 256; return run_h<4, 4, 8, 2>(a, st->p, s);
    case 7: *tile_px = 256; return run_h<4, 4, 8, 3>(a, st->p, s);
    case 8: *tile_px = 256; return run_h<4, 8, 4, 3>(a, st->p, s);
    case 9: *tile_px = 256; return run_h<4, 8, 4, 4>(a, st->p, s);
    case 10: *tile_px = 512; return run_h<8, 4, 2, 3>(a, st->p, s);
    case 11: *tile_px = 512; return run_h<8, 4, 2, 4>(a, st->p, s);
    case 12: *tile_px = 512; return run_h<8, 4, 4, 3>(a, st->p, s);
    case 13: *tile_px = 128; return run_h<2, 8, 8, 3>(a, st->p, s);
    case 14: *tile_px = 256; return run_h<4, 4, 4, 6>(a, st->p, s);
    case 15: *tile_px = 256; return run_h<4, 4, 8, 0>(a, st->p, s);
  }
  return -1;
}

const char* tune_w_name(int v) {
  static const char* names[] = {"up4 nb0", "up4 nb2", "up4 nb3", "up4 nb4", "up2 nb4", "up2 nb6", "up2 nb8", "up8 nb2"};
  return (v >= 0 && v < 8) ? names[v] : nullptr;
}

// nblk pixel blocks (a_slab must hold nblk slabs)
int tune_w(const espm_mu_state* st, int v, int nblk, void* stream) {
  WAccumArgs a = make_w_args(st);
  a.ppb = (st->p + nblk - 1) / nblk;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (v) {
    case 0: return run_w<4, 0>(a, nblk, s);
    case 1: return run_w<4, 2>(a, nblk, s);
    case 2: return run_w<4, 3>(a, nblk, s);
    case 3: return run_w<4, 4>(a, nblk, s);
    case 4: return run_w<2, 4>(a, nblk, s);
    case 5: return run_w<2, 6>(a, nblk, s);
    case 6: return run_w<2, 8>(a, nblk, s);
    case 7: return run_w<8, 2>(a, nblk, s);
  }
  return -1;
}

}  // extern "C"
