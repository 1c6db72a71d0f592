// One launch for both half-steps of an iteration on the sparse count store (x_dtype = ESPM_X_ELL, 512-pixel H tiles).
//
// The W accumulation of a pixel block needs the new H of THAT block only (updates.py:38-39, :53-59: R = X / (GW H') and
// R H'^T are sums over pixels), so the workgroup that has just updated the 1024 pixels of a block can go on and walk the
// block's channel lists without waiting for anybody else: no grid-wide dependency sits between the two half-steps.
// What IS global - the row sums of H' in the denominator of W', the sum of the slabs - stays in the reduction launch
// that follows (w_reduce_update_kernel / w_reduce_kernel, mu_w_step.hip).
//
//   workgroup = 16 waves = one block of ESPM_ELL_PB = 1024 pixels = two H tiles of 512 pixels
//   1. GW table -> LDS (address 0), as in h_step_ell_kernel
//   2. H walk: waves 0..7 tile 2b, waves 8..15 tile 2b + 1, each half exactly as h_step_ell_kernel walks its window
//      (pairs of list groups, two partial numerators per pixel)
//   3. epilogue, one thread per pixel (h_epilogue): H' -> h[1 - src] (the next iteration's stencil and rel_H need it in
//      memory) and, instead of the transposed copy h_t in memory, straight into the LDS table of the W walk, which
//      takes the place of the GW table (address 0: the unit entries of both list sets address their table without a base)
//   4. W walk: w_accum_ell_kernel's body; the GW row of a lane's channel comes from gw_s (L2)
//
// Against the two launches this saves a kernel boundary, the table prologue of the W accumulation (1024 rows of h_t
// from memory, a barrier), and 2 x 4 KP p bytes of h_t traffic.  The block's record goes to the hpart slot of its first
// tile, zeros to the slot of the second (HStepArgs::rec_nb), so the readers of the records do not change.
#pragma once
#include "mu_ell_kernel.hpp"

namespace espm {

struct FusedArgs {
  HStepArgs h;      // write_h = 1, ell_tp = 512; h_t unused
  WAccumArgs w;     // h_t unused
};

template <int K, bool LOSS, int UNR_H, int UNR_W>
__global__ __launch_bounds__(ESPM_ELL_WTHREADS) void mu_fused_ell_kernel(const FusedArgs fa) {
  constexpr int NT = ESPM_ELL_WTHREADS;       // 1024 threads
  constexpr int TP = ESPM_ELL_TILE;           // 512 pixels per half
  constexpr int PB = ESPM_ELL_PB;             // 1024 pixels per workgroup
  static_assert(PB == 2 * TP && NT == PB, "a workgroup is two H tiles and one thread per pixel");
  constexpr bool PAIRS_OK = K <= ESPM_ELL_PAIR_MAX_K;
  constexpr int NPARTS = PAIRS_OK ? 2 : 1;
  const HStepArgs& a = fa.h;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;   // [n_pad] rows of GW, later [PB] rows of H'
  ell_table_at_lds_zero(tab);
  const int tab_rows = a.n_pad > PB ? a.n_pad : PB;
  float* part = smem + (size_t)tab_rows * EllTab<K>::FLOATS;   // [NPARTS][K][PB] numerators (pixel = its place in the block), then reduction scratch
  if (a.tail_on && blockIdx.x == gridDim.x - 1) {   // (uniform) the extra workgroup: tail of the previous W update
    w_tail_body<10>(a.tail, reinterpret_cast<double*>(smem));
    return;
  }
  double* cs_lds = a.cs_parts ? reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(smem) + a.cs_lds_off) : nullptr;
  if (cs_lds && (int)(threadIdx.x >> 6) < K) {      // wave kk: column sum kk of G W' from the W update's partials
    const int kk = threadIdx.x >> 6;
    double v = 0.0;
    for (int j = threadIdx.x & 63; j < a.cs_nbk; j += 64) v += a.cs_parts[(size_t)kk * a.cs_nbk + j];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) cs_lds[kk] = v;
  }
  for (int r = threadIdx.x; r < a.n_pad; r += NT) {
    const float4* src = reinterpret_cast<const float4*>(a.gw_s + (size_t)r * KP);
    EllTab<K>::put(tab, a.n_pad, r, src[0], src[1]);
  }
  if constexpr (PAIRS_OK) {  // second partial numerator: only the pixels of the longer group of a pair receive one
#pragma unroll
    for (int kk = 0; kk < K; ++kk) part[((size_t)K + kk) * PB + threadIdx.x] = 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave >> 3, hw = wave & 7;          // H tile of this wave, its number inside the tile's eight
  const int blk0 = blockIdx.x * PB;
  const int tile0 = blk0 + half * TP;
  const bool tile_ok = tile0 < a.p_pad;               // (an odd number of tiles: the last block has one)
  float kl = 0.f;

  // rows [x0, x1) of list group gi of the tile -> partial numerator `slot` of its pixels
  auto walk_rows = [&](int gi, int x0, int x1, int slot) {
    const int grp = tile0 / 64 + gi;
    const int lp = a.ell_pix[tile0 + gi * 64 + lane];   // slot -> pixel of the window (lists ordered by length)
    const int px = tile0 + lp;
    float hk[K], acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      hk[kk] = a.h_in[(size_t)kk * a.p_pad + px];
      acc[kk] = 0.f;
    }
    const int beg = a.ell_off[2 * grp], mid = a.ell_off[2 * grp + 1] - beg;
    const uint32_t* lrow = a.ell + (size_t)beg * 64 + lane;
    if (x0 < mid) {
      ell_walk<K, UNR_H>(lrow + (size_t)x0 * 64, min(x1, mid) - x0, EllGetUnit<K>(a.n_pad), [&](float, const float (&g)[K]) {
        const float r = __builtin_amdgcn_rcpf(ell_dot<K>(g, hk));
        ell_axpy<K>(acc, g, r);
        if constexpr (LOSS) kl += __builtin_amdgcn_logf(r);
      });
    }
    if (x1 > mid) {
      const int g0 = max(x0, mid);
      ell_walk<K, UNR_H>(lrow + (size_t)g0 * 64, x1 - g0, EllGet<K>(tab, a.n_pad, a.ell_bits), [&](float x, const float (&g)[K]) {
        const float y = ell_dot<K>(g, hk);
        const float r = LOSS ? fmaf(x, __builtin_amdgcn_rcpf(y), 1e-37f) : x * __builtin_amdgcn_rcpf(y);
        ell_axpy<K>(acc, g, r);
        if constexpr (LOSS) kl = fmaf(x, __builtin_amdgcn_logf(r), kl);
      });
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) part[((size_t)slot * K + kk) * PB + half * TP + lp] = acc[kk];
    if (LOSS && slot == 0) kl += fmaxf(a.ell_klc[px], 0.f);
  };
  auto group_rows = [&](int gi) { return a.ell_off[2 * (tile0 / 64 + gi) + 2] - a.ell_off[2 * (tile0 / 64 + gi)]; };

  if (tile_ok) {
    if constexpr (PAIRS_OK) {
      const int gl = hw < 4 ? hw : 7 - hw;           // the longer group of this wave's pair
      const int len_l = group_rows(gl), len_s = group_rows(7 - gl);
      const int hrows = min(len_l, (len_l + len_s + 1) / 2);
      if (hw < 4) {
        walk_rows(gl, 0, hrows, 0);
      } else {
        walk_rows(hw, 0, len_s, 0);
        if (hrows < len_l) walk_rows(gl, hrows, len_l, 1);
      }
    } else {
      walk_rows(hw, 0, group_rows(hw), 0);
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < K; ++kk) part[(size_t)kk * PB + threadIdx.x] = 0.f;   // (no tile: its pixels lie beyond p, never read)
  }
  // per-pixel epilogue over the 1024 pixels of the block; H' rows go into the LDS table of the W walk (rows of the
  // pixels beyond p: ones, never referenced by an entry with a count)
  h_epilogue<K, true, 0>(a, part, NPARTS, PB, blk0, LOSS ? kl : 0.f, cs_lds, tab, PB);

  // ---- W accumulation over the block's channel lists (w_accum_ell_kernel's body with csplit = 1) ----
  const WAccumArgs& w = fa.w;
  const int b = blockIdx.x;
  constexpr int nw = NT / 64;
  for (int t = 0; t * nw < w.n_cg; ++t) {
    const int cg = t * nw + ((t & 1) ? nw - 1 - wave : wave);
    if (cg >= w.n_cg) continue;
    const int c = w.chan_perm[((size_t)b * w.n_cg + cg) * 64 + lane];
    const float* gsrc = w.gw_s + (size_t)(c < 0 ? 0 : c) * KP;
    float gw[K], acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      gw[kk] = gsrc[kk];
      acc[kk] = 0.f;
    }
    const int32_t* off = w.ell_off + 2 * ((size_t)b * w.n_cg + cg);
    const int beg = off[0], mid = off[1], end = off[2];
    const uint32_t* lrow = w.ell + (size_t)beg * 64 + lane;
    ell_walk<K, UNR_W>(lrow, mid - beg, EllGetUnit<K>(PB), [&](float, const float (&h)[K]) {
      ell_axpy<K>(acc, h, __builtin_amdgcn_rcpf(ell_dot<K>(h, gw)));
    });
    ell_walk<K, UNR_W>(lrow + (size_t)(mid - beg) * 64, end - mid, EllGet<K>(tab, PB, ESPM_ELL_PBITS), [&](float x, const float (&h)[K]) {
      const float r = x * __builtin_amdgcn_rcpf(ell_dot<K>(h, gw));
      ell_axpy<K>(acc, h, r);
    });
    if (c >= 0) {
#pragma unroll
      for (int kk = 0; kk < K; ++kk) w.a_slab[((size_t)b * K + kk) * w.n_pad + c] = acc[kk];
    }
  }
}

}  // namespace espm
