// extern "C" surface of libespm_mu.so (declared in include/espm_mu.h): argument checking, kernel
// selection and launch sequencing.  No allocation, no host synchronisation.
#include <vector>
#include <stdarg.h>
#include <stdio.h>

#include "mu_common.hpp"
#include "mu_xchg.hpp"

namespace espm {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return ESPM_OK;
  return set_error(ESPM_EHIP, "%s: %s", what, hipGetErrorString(e));
}

static int roundup(int v, int m) { return (v + m - 1) / m * m; }

// every field of espm_mu_state, in declaration order (tests compare espm_mu_state_layout() with the header: a field missing here fails them)
#define ESPM_MU_STATE_FIELDS(F) \
  F(struct_size) F(abi_version) F(n) F(m) F(k) F(p) F(nx) F(ny) F(n_pad) F(p_pad) F(x_dtype) F(tile_px) F(nblk_w) \
  F(x_tile) F(n_cm) F(h_variant) F(p_total) F(simplex_h) F(simplex_w) F(grid_mode) F(compute_loss) F(lambda_l) \
  F(sigma_l) F(eps_reg) F(log_shift) F(dicotomy_tol) F(rel_tol) F(xscale) F(gw_floor) F(x_cm) F(x_pm) F(g) \
  F(colsum_g) F(w) F(gw_s) F(colsum_gw) F(gw_a) F(gw_p) F(h) F(h_t) F(mu) F(fixed_h) F(fixed_w) F(simplex_rows) \
  F(halo_top) F(halo_bot) F(hpart) F(hstat) F(a_slab) F(a) F(w_scratch) F(hist) F(hist_len) F(cur) F(it) F(ell_h) \
  F(ell_h_off) F(ell_klc) F(ell_w) F(ell_w_off) F(chan_perm) F(ell_cbits) F(n_cg) F(pix_perm) F(g_t) F(breg_sr_px) \
  F(breg_sr_ch) F(h_rule) F(pg_gamma_w) F(pg_q) F(ell_fill_px) F(ell_fill_num) F(ell_fill_n) F(tail_mode) F(no_fused) \
  F(ell_pb) F(ell_stream)

// the caller's view of the state must be this library's (include/espm_mu.h, ESPM_MU_ABI_VERSION): checked before any field is read
static int check_abi(const espm_mu_state* st) {
  ESPM_REQUIRE(st != nullptr, "state is NULL");
  ESPM_REQUIRE(st->struct_size == (uint32_t)sizeof(espm_mu_state) && st->abi_version == (uint32_t)ESPM_MU_ABI_VERSION,
               "espm_mu_state of %u bytes, ABI %u handed to a library built for %zu bytes, ABI %d: the binding's copy of the layout has drifted",
               st->struct_size, st->abi_version, sizeof(espm_mu_state), ESPM_MU_ABI_VERSION);
  return ESPM_OK;
}

static int check_state(const espm_mu_state* st) {
  if (int rc = check_abi(st)) return rc;
  ESPM_REQUIRE(st->n >= 1 && st->p >= 1, "n=%d, p=%d must be >= 1", st->n, st->p);
  if (st->k < ESPM_MIN_K || st->k > ESPM_MAX_K)
    return set_error(ESPM_EUNSUPPORTED, "k=%d: this build supports %d..%d components", st->k, ESPM_MIN_K, ESPM_MAX_K);
  ESPM_REQUIRE(st->n_pad == roundup(st->n, ESPM_NPAD) && st->p_pad == roundup(st->p, ESPM_PPAD),
               "n_pad/p_pad (%d, %d) do not match n, p (%d, %d); call espm_mu_query", st->n_pad, st->p_pad, st->n,
               st->p);
  ESPM_REQUIRE(st->x_dtype >= ESPM_X_F32 && st->x_dtype <= ESPM_X_ELL, "bad x_dtype %d", st->x_dtype);
  if (st->x_dtype == ESPM_X_ELL) {
    ESPM_REQUIRE(st->ell_h && st->ell_h_off && st->ell_klc && st->ell_w && st->ell_w_off && st->chan_perm && st->pix_perm,
                 "the sparse count store needs ell_h, ell_h_off, ell_klc, ell_w, ell_w_off, chan_perm and pix_perm");
    ESPM_REQUIRE((st->tile_px == 64 || st->tile_px == 128 || st->tile_px == 256 || st->tile_px == ESPM_ELL_TILE) &&
                     st->ell_pb == 2 * st->tile_px && st->nblk_w == (st->p + st->ell_pb - 1) / st->ell_pb &&
                     st->n_cg == (st->n + 63) / 64 && st->h_variant == 0,
                 "sparse count store: tile_px must be 64..%d, ell_pb 2 tile_px, nblk_w ceil(p / ell_pb), n_cg ceil(n / 64); call espm_mu_query",
                 ESPM_ELL_TILE);
  }
  ESPM_REQUIRE(st->ell_stream == 0 || (st->ell_stream == 1 && st->x_dtype == ESPM_X_ELL), "ell_stream=%d: 0, or 1 with the sparse store", st->ell_stream);
  ESPM_REQUIRE(st->ell_fill_n >= 0 && (st->ell_fill_n == 0 || (st->x_dtype == ESPM_X_ELL && st->ell_fill_px && st->ell_fill_num)),
               "ell_fill_n=%d needs the sparse store, ell_fill_px and ell_fill_num", st->ell_fill_n);
  ESPM_REQUIRE(st->grid_mode == 0 || (st->nx >= 1 && st->ny >= 1 && st->nx * st->ny == st->p),
               "grid %d x %d does not match p=%d", st->nx, st->ny, st->p);
  ESPM_REQUIRE(st->xscale > 0.f, "xscale must be positive");
  ESPM_REQUIRE(st->n_cm == roundup(st->n, ESPM_NCM), "n_cm must be roundup(n, %d)", ESPM_NCM);
  ESPM_REQUIRE((st->breg_sr_px == nullptr) == (st->breg_sr_ch == nullptr), "breg_sr_px and breg_sr_ch come together");
  ESPM_REQUIRE(!st->breg_sr_px || (st->m == 0 && !st->simplex_w && st->n <= 4096),
               "the Bregman variant is built for G = identity without simplex_W (and n <= 4096)");
  ESPM_REQUIRE(st->h_rule == 0 || ((st->h_rule == 1 || st->h_rule == 2) && !st->breg_sr_px), "h_rule must be 0, 1 or 2 (1, 2 not with the Bregman variant)");
  ESPM_REQUIRE(!(st->pg_gamma_w > 0.f) || (!st->simplex_w && !st->breg_sr_px && (st->m == 0 || (long)st->m * st->k <= 8192)),
               "the projected-gradient W step has no simplex over W (updates.py:368-369), no Bregman variant, and needs M k <= 8192 with a dictionary");
  ESPM_REQUIRE(st->h_variant == 0, "h_variant %d is not built (the matrix-core H-step was retired, DESIGN.md)", st->h_variant);
  ESPM_REQUIRE(st->x_tile >= 64 && ESPM_PPAD % st->x_tile == 0 && st->x_tile % st->tile_px == 0,
               "x_tile=%d must divide %d and be a multiple of tile_px=%d", st->x_tile, ESPM_PPAD, st->tile_px);
  ESPM_REQUIRE(st->m >= 0, "m must be >= 0");
  ESPM_REQUIRE(st->m == 0 || (st->g && st->colsum_g), "m=%d needs g and colsum_g", st->m);
  if (st->log_shift > 0.f && st->simplex_h && (double)st->k * (double)st->log_shift >= 1.0)
    return set_error(ESPM_ENOSOLUTION, "No solution exists! (k * log_shift >= 1)");
  return ESPM_OK;
}

static int nblk_h(const espm_mu_state* st) { return (st->p + st->tile_px - 1) / st->tile_px; }

// Both half-steps in one launch (mu_fused_kernel.hpp): sparse store, the default H rule, LDS for the table and the numerators
// of a block of ell_pb pixels.  One record per pixel BLOCK.  (Round 2 kept the two launches below blocks of 512 pixels - images
// under 2^17 pixels, shards - where one workgroup per CU left the CU idle through its serial phases.  With those phases cut
// in round 3 - prologue loads issued together, the update's inputs staged by the prologue, one barrier around the record
// reduction, no extra workgroup for the W update's tail - the fused launch wins at every block size: a 64-row shard 47.4 -> 42.8 us
// per iteration, profiles/r03d_shard_iter*.log - where the blocks cover the chip (ESPM_FUSED_MIN_BLOCKS); an image of fewer, small
// blocks keeps the two launches, whose smaller workgroups spread over more CUs.)
static bool fused_ok(const espm_mu_state* st) {
  return ESPM_MIN_K <= 8 && st->x_dtype == ESPM_X_ELL && st->h_rule == 0 && st->no_fused != 1 &&
         (st->ell_pb >= ESPM_FUSED_MIN_PB || st->nblk_w >= ESPM_FUSED_MIN_BLOCKS || st->no_fused == 3) &&
         fused_ell_lds_bytes(st->n_pad, st->k, st->ell_pb) <= ESPM_ELL_LDS_MAX;
}
static HStepArgs fused_h_args(const espm_mu_state* st, int src) {
  HStepArgs a = make_h_args(st, src, 1);
  a.rec_nb = nblk_h(st);
  return a;
}

}  // namespace espm

using namespace espm;

extern "C" {

const char* espm_mu_version(void) { return "espm_mu 0.3 (gfx950)"; }
const char* espm_mu_last_error(void) { return g_err; }

size_t espm_mu_state_size(void) { return sizeof(espm_mu_state); }
int espm_mu_abi_version(void) { return ESPM_MU_ABI_VERSION; }

const char* espm_mu_state_layout(void) {
  static char text[4096];
  static bool done = false;
  if (!done) {
    size_t len = 0;
#define F(name)                                                                                                      \
  len += (size_t)snprintf(text + len, sizeof(text) - len, #name ":%zu:%zu;", offsetof(espm_mu_state, name), \
                          sizeof(((espm_mu_state*)nullptr)->name));
    ESPM_MU_STATE_FIELDS(F)
#undef F
    done = true;
  }
  return text;
}

int espm_mu_query(espm_mu_state* st) {
  if (int rc = check_abi(st)) return rc;
  ESPM_REQUIRE(st->n >= 1 && st->p >= 1, "n=%d, p=%d must be >= 1", st->n, st->p);
  st->n_pad = roundup(st->n, ESPM_NPAD);
  st->p_pad = roundup(st->p, ESPM_PPAD);
  int cus = 256;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  // H-step: 256-pixel tiles (4 waves) when that gives >= 2 workgroups per CU, else 128-pixel tiles
  // with 8 waves splitting the channel range (tuned on MI355X, tools/tune).
  // (the widest build, 17..32 components: 128-pixel tiles only - the partial numerators of a 256-pixel tile would take 68..128 KB of LDS)
  const int big = 256;
  st->tile_px = (ESPM_KP <= 16 && (st->p + big - 1) / big >= 2 * cus) ? big : 128;
  st->x_tile = st->tile_px;
  st->n_cm = roundup(st->n, ESPM_NCM);
  st->h_variant = 0;
  st->n_cg = (st->n + 63) / 64;
  st->ell_pb = 0;
  if (st->x_dtype == ESPM_X_ELL) {  // sparse count store: fixed decomposition (mu_ell_kernel.hpp)
    // H-step: 512 pixels per workgroup (8 waves, one 64-pixel list group each); smaller images split every group
    // over 2, 4 or 8 waves so that about two workgroups per CU remain
    st->tile_px = ESPM_ELL_TILE;
    while (st->tile_px > 64 && (st->p + st->tile_px - 1) / st->tile_px < 2 * cus) st->tile_px /= 2;
    st->x_tile = st->tile_px;
    // W accumulation: blocks of two H tiles (1024 pixels for an image that fills the chip; about one block per CU below that,
    // so that one workgroup can own a block through both half-steps: mu_fused_kernel.hpp)
    st->ell_pb = 2 * st->tile_px;
    st->nblk_w = (st->p + st->ell_pb - 1) / st->ell_pb;
    st->ell_cbits = 1;
    while ((1 << st->ell_cbits) < st->n) ++st->ell_cbits;
    if (st->ell_cbits > 14) return set_error(ESPM_EUNSUPPORTED, "sparse count store: n=%d needs more than 14 index bits", st->n);
    return ESPM_OK;
  }
  // W accumulation: about 2 workgroups per CU, at least 16 pixels each.
  const int ychunks = (st->x_dtype != ESPM_X_F32 && ESPM_MIN_K <= 8) ? (st->n_pad + 2047) / 2048 : (st->n_pad + 1023) / 1024;
  int target = (2 * cus + ychunks - 1) / ychunks;
  int by_px = (st->p + 15) / 16;
  int nb = target < by_px ? target : by_px;
  if (nb < 1) nb = 1;
  const int ppb = (st->p + nb - 1) / nb;
  st->nblk_w = (st->p + ppb - 1) / ppb;
  return ESPM_OK;
}

int espm_mu_pack_x(const void* src, int src_dtype, int src_layout, int64_t ld, int n, int p, void* x_cm, void* x_pm,
                   int x_dtype, int n_pad, int p_pad, int x_tile, int n_cm, espm_stream_t stream) {
  ESPM_REQUIRE(src && x_pm, "pack_x: NULL pointer");  // x_cm may be NULL (sparse store ingest needs x_pm only)
  ESPM_REQUIRE(n >= 1 && p >= 1 && n_pad == roundup(n, ESPM_NPAD) && p_pad == roundup(p, ESPM_PPAD),
               "pack_x: bad shape n=%d p=%d n_pad=%d p_pad=%d", n, p, n_pad, p_pad);
  ESPM_REQUIRE(src_dtype == ESPM_SRC_F32 || src_dtype == ESPM_SRC_F64, "pack_x: bad src_dtype %d", src_dtype);
  ESPM_REQUIRE(src_layout == ESPM_LAYOUT_CM || src_layout == ESPM_LAYOUT_PM, "pack_x: bad layout %d", src_layout);
  ESPM_REQUIRE(ld >= (src_layout == ESPM_LAYOUT_CM ? p : n), "pack_x: leading dimension too small");
  ESPM_REQUIRE(x_tile >= 64 && ESPM_PPAD % x_tile == 0, "pack_x: x_tile %d must divide %d", x_tile, ESPM_PPAD);
  ESPM_REQUIRE(n_cm == roundup(n, ESPM_NCM), "pack_x: n_cm must be roundup(n, %d)", ESPM_NCM);
  return launch_pack_x(src, src_dtype, src_layout, ld, n, p, x_cm, x_pm, x_dtype, n_pad, p_pad, x_tile, n_cm,
                       static_cast<hipStream_t>(stream));
}

static int check_ell_geometry(const espm_mu_state* st) {
  if (int rc = check_abi(st)) return rc;
  ESPM_REQUIRE(st->x_dtype == ESPM_X_ELL && st->n >= 1 && st->p >= 1 && st->n_pad == roundup(st->n, ESPM_NPAD) &&
                   st->p_pad == roundup(st->p, ESPM_PPAD) && st->n_cg == (st->n + 63) / 64 &&
                   st->ell_pb == 2 * st->tile_px && st->nblk_w == (st->p + st->ell_pb - 1) / st->ell_pb && st->ell_cbits >= 1 &&
                   st->ell_cbits <= 14 && (1 << st->ell_cbits) >= st->n &&
                   (st->tile_px == 64 || st->tile_px == 128 || st->tile_px == 256 || st->tile_px == ESPM_ELL_TILE),
               "sparse store builder: set x_dtype = ESPM_X_ELL and call espm_mu_query first");
  return ESPM_OK;
}

int espm_mu_ell_count_hist(const espm_mu_state* st, const void* x_pm_u8, int32_t* cnt_px, int32_t* cnt_bc, float* ell_klc, uint8_t* bkt_px,
                           uint8_t* bkt_bc, espm_stream_t stream) {
  if (int rc = check_ell_geometry(st)) return rc;
  ESPM_REQUIRE(x_pm_u8 && cnt_px && cnt_bc && ell_klc, "ell_count: NULL pointer");
  ESPM_REQUIRE((bkt_px == nullptr) == (bkt_bc == nullptr), "ell_count: the two histograms come together");
  return launch_ell_count(static_cast<const uint8_t*>(x_pm_u8), st->n, st->n_pad, st->p, st->p_pad, st->ell_cbits, st->n_cg,
                          st->nblk_w, st->ell_pb, cnt_px, cnt_bc, ell_klc, static_cast<hipStream_t>(stream), bkt_px, bkt_bc);
}

int espm_mu_ell_count(const espm_mu_state* st, const void* x_pm_u8, int32_t* cnt_px, int32_t* cnt_bc, float* ell_klc,
                      espm_stream_t stream) {
  return espm_mu_ell_count_hist(st, x_pm_u8, cnt_px, cnt_bc, ell_klc, nullptr, nullptr, stream);
}

int espm_mu_ell_plan(const espm_mu_state* st, const int32_t* cnt_px, const int32_t* cnt_bc, int32_t* chan_perm,
                     int32_t* pix_perm, int32_t* ell_h_off, int32_t* ell_w_off, int64_t* rows, espm_stream_t stream) {
  if (int rc = check_ell_geometry(st)) return rc;
  ESPM_REQUIRE(cnt_px && cnt_bc && chan_perm && pix_perm && ell_h_off && ell_w_off && rows, "ell_plan: NULL pointer");
  return launch_ell_plan(cnt_px, cnt_bc, st->n, st->n_cg, st->nblk_w, st->p_pad, st->tile_px, chan_perm, pix_perm, ell_h_off,
                         ell_w_off, reinterpret_cast<long long*>(rows), static_cast<hipStream_t>(stream));
}

int espm_mu_ell_fill(const espm_mu_state* st, const void* x_pm_u8, const int32_t* chan_perm, const int32_t* pix_perm,
                     const int32_t* ell_h_off, const int32_t* ell_w_off, uint32_t* ell_h, uint32_t* ell_w,
                     espm_stream_t stream) {
  return espm_mu_ell_fill_hist(st, x_pm_u8, chan_perm, pix_perm, ell_h_off, ell_w_off, ell_h, ell_w, nullptr, nullptr, stream);
}

int espm_mu_ell_fill_hist(const espm_mu_state* st, const void* x_pm_u8, const int32_t* chan_perm, const int32_t* pix_perm,
                          const int32_t* ell_h_off, const int32_t* ell_w_off, uint32_t* ell_h, uint32_t* ell_w, const uint8_t* bkt_px,
                          const uint8_t* bkt_bc, espm_stream_t stream) {
  if (int rc = check_ell_geometry(st)) return rc;
  ESPM_REQUIRE((bkt_px == nullptr) == (bkt_bc == nullptr), "ell_fill: the two histograms come together");
  ESPM_REQUIRE(x_pm_u8 && chan_perm && pix_perm && ell_h_off && ell_w_off && ell_h && ell_w, "ell_fill: NULL pointer");
  // st->x_cm, if set: the 8-bit counts once more, channel-major in tiles of ESPM_PPAD pixels (include/espm_mu.h) - the channel lists are filled from it
  ESPM_REQUIRE(!st->x_cm || st->n_cm == roundup(st->n, ESPM_NCM), "ell_fill: x_cm is set but n_cm=%d is not n rounded up to %d", st->n_cm, ESPM_NCM);
  return launch_ell_fill(static_cast<const uint8_t*>(x_pm_u8), st->n, st->n_pad, st->p, st->p_pad, st->ell_cbits, st->n_cg,
                         st->nblk_w, st->tile_px, st->ell_pb, chan_perm, pix_perm, ell_h_off, ell_w_off, ell_h, ell_w,
                         static_cast<hipStream_t>(stream), static_cast<const uint8_t*>(st->x_cm), st->n_cm, bkt_px, bkt_bc);
}

int espm_mu_hstat(const espm_mu_state* st, int which, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(which == 0 || which == 1, "which must be 0/1");
  return launch_hstat(st->h[which], st->k, st->p, st->p_pad, st->hstat[which], static_cast<hipStream_t>(stream));
}

// W' needs nothing global but the row sums of the new H: the reduction workgroups finish W themselves
static bool w_update_is_local(const espm_mu_state* st) {
  return st->m == 0 && !st->simplex_w && st->n >= 64 && st->w_scratch != nullptr;
}
// The simplex over W with G = identity and every row in the simplex: the multipliers follow from per-component sums over the
// channels, so the update runs as two many-workgroup launches (slab reduction + those sums, then w_simplex_update_kernel)
// instead of the reduction and ONE workgroup that finds the multipliers (mu_w_step.hip).  Its tail is the local update's.
static bool w_simplex_split(const espm_mu_state* st) {
  return st->m == 0 && st->simplex_w && !st->simplex_rows && !st->breg_sr_ch && !(st->pg_gamma_w > 0.f) && st->n >= 64 &&
         st->w_scratch != nullptr && st->n_pad % 32 == 0 && st->no_fused != 1;
}
// where the partials of the bracket live: behind the tail's three arrays of k * nbk doubles in w_scratch (2 n k floats)
static double* simplex_bparts(const espm_mu_state* st) {
  return reinterpret_cast<double*>(st->w_scratch) + 3 * (size_t)st->k * ((st->n_pad + 31) / 32);
}
// the two launches: st->a_slab -> st->a and the partials; then W', G W' and the tail's partials
static int w_simplex_reduce_update(const espm_mu_state* st, int src, int slot, const HFinalizeArgs* fin, WTailArgs* defer_tail, hipStream_t s);

static WFinishArgs finish_args(const espm_mu_state* st, int src, int hsrc, int slot, int update_w) {
  WFinishArgs a;
  a.g = st->m > 0 ? st->g : nullptr;
  a.g_t = st->m > 0 ? st->g_t : nullptr;
  a.colsum_g = st->colsum_g;
  a.w_old = st->w[src];
  a.w_new = update_w ? st->w[1 - src] : st->w[src];
  a.a = st->a;
  a.hstat = st->hstat[hsrc];
  a.fixed_w = st->fixed_w;
  a.simplex_rows = st->simplex_rows;
  a.scratch = st->w_scratch;
  a.breg_sr = st->breg_sr_ch;
  a.pg_gamma_w = st->pg_gamma_w;
  a.pg_q = (st->pg_q && st->pg_gamma_w > 0.f && slot >= 0) ? st->pg_q + 2 * (size_t)slot + 1 : nullptr;
  a.gw_s = st->gw_s;
  a.colsum_gw = st->colsum_gw;
  a.gw_a = nullptr;  // reserved (matrix-core H-step, retired)
  a.gw_p = nullptr;
  a.n_cm = st->n_cm;
  a.hist_slot = (slot >= 0 && st->hist) ? st->hist + (size_t)slot * ESPM_HI_STRIDE : nullptr;
  a.n = st->n;
  a.m = st->m;
  a.k = st->k;
  a.n_pad = st->n_pad;
  a.simplex_w = st->simplex_w;
  a.update_w = update_w;
  a.log_shift = st->log_shift;
  // (the reference's W step has no tolerance argument: it calls dichotomy_simplex with the module constant,
  //  updates.py:61-68 / conf.py - only the H step takes the estimator's dicotomy_tol)
  a.tol = ESPM_W_DICOTOMY_TOL;
  a.rel_tol = st->rel_tol;
  a.xscale = st->xscale;
  a.gw_floor = st->gw_floor;
  return a;
}

static int w_simplex_reduce_update(const espm_mu_state* st, int src, int slot, const HFinalizeArgs* fin, WTailArgs* defer_tail, hipStream_t s) {
  if (st->log_shift > 0.f && (double)st->n * (double)st->log_shift >= 1.0)
    return set_error(ESPM_ENOSOLUTION, "No solution exists! (rows * log_shift >= 1)");
  double* bparts = simplex_bparts(st);
  if (int rc = launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, fin, s, st->w[src], bparts, st->n, st->k, st->n_pad)) return rc;
  return launch_w_simplex_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a, bparts, (double)ESPM_W_DICOTOMY_TOL, s, defer_tail);
}

int espm_mu_build_gw(const espm_mu_state* st, int which, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(which == 0 || which == 1, "which must be 0/1");
  return launch_w_finish(finish_args(st, which, 0, -1, 0), static_cast<hipStream_t>(stream));
}

int espm_mu_step_h(const espm_mu_state* st, int src, int write_h, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  if (st->x_dtype == ESPM_X_ELL) {
    HStepArgs a = make_h_args(st, src, write_h);
    if (st->tail_mode & ESPM_TAIL_RIDE) {  // the tail of the W update that produced this state rides along (include/espm_mu.h)
      ESPM_REQUIRE(st->it >= 1 && (w_update_is_local(st) || w_simplex_split(st)), "tail_mode: no local W update produced state %d", st->it);
      a.tail = make_w_tail_args(finish_args(st, 1 - src, src, st->it, 1));
      a.cs_parts = a.tail.parts;
      a.cs_nbk = a.tail.nbk;
      a.tail_on = 1;
    }
    if (a.fill_num) {  // pixels without counts: the numerator of their log_shift fill first (include/espm_mu.h)
      if (int rc = launch_ell_fill_num(st->gw_s, st->h[src], st->ell_fill_px, st->ell_fill_n, st->n, st->k, st->p_pad, st->log_shift,
                                       st->ell_fill_num, static_cast<hipStream_t>(stream)))
        return rc;
    }
    return launch_h_ell(a, nblk_h(st), static_cast<hipStream_t>(stream));
  }
  return dispatch_h_step(make_h_args(st, src, write_h), st->x_dtype, st->tile_px, nblk_h(st),
                         static_cast<hipStream_t>(stream));
}

static HFinalizeArgs finalize_args(const espm_mu_state* st, int src, int slot, bool write_hstat) {
  HFinalizeArgs a;
  a.hpart = st->hpart;
  a.colsum_gw = st->colsum_gw;
  a.hstat_in = st->hstat[src];
  a.hstat_out = write_hstat ? st->hstat[1 - src] : nullptr;
  a.hist_slot = st->hist + (size_t)slot * ESPM_HI_STRIDE;
  a.nblk = nblk_h(st);
  a.k = st->k;
  a.compute_loss = st->compute_loss;
  a.have_prev = st->it > 0;
  a.xscale = st->xscale;
  a.pg_q = (st->pg_q && st->h_rule == 2 && write_hstat) ? st->pg_q + 2 * (size_t)slot : nullptr;   // (not for loss-only evaluations)
  return a;
}

int espm_mu_h_finalize(const espm_mu_state* st, int src, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  ESPM_REQUIRE(slot >= 0 && slot < st->hist_len, "history slot %d outside [0, %d)", slot, st->hist_len);
  return launch_h_finalize(finalize_args(st, src, slot, true), static_cast<hipStream_t>(stream));
}

/* loss-only variant: does not touch hstat[1-src] */
static int h_finalize_loss_only(const espm_mu_state* st, int src, int slot, hipStream_t stream) {
  return launch_h_finalize(finalize_args(st, src, slot, false), stream);
}

int espm_mu_loss_only(const espm_mu_state* st, int src, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(slot >= 0 && slot < st->hist_len, "history slot %d outside [0, %d)", slot, st->hist_len);
  if (int rc = espm_mu_step_h(st, src, 0, stream)) return rc;
  return h_finalize_loss_only(st, src, slot, static_cast<hipStream_t>(stream));
}

int espm_mu_w_accum(const espm_mu_state* st, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(st->nblk_w >= 1, "nblk_w must be >= 1");
  if (st->x_dtype == ESPM_X_ELL) return launch_w_ell(make_w_args(st), st->k, st->nblk_w, static_cast<hipStream_t>(stream));
  return dispatch_w_accum(make_w_args(st), st->k, st->x_dtype, st->nblk_w, static_cast<hipStream_t>(stream));
}

int espm_mu_fused_applies(const espm_mu_state* st) {
  if (check_state(st)) return 0;
  return fused_ok(st) ? 1 : 0;
}

int espm_mu_step_hw(const espm_mu_state* st, int src, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  if (!fused_ok(st)) {
    if (int rc = espm_mu_step_h(st, src, 1, stream)) return rc;
    return espm_mu_w_accum(st, stream);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  HStepArgs a = fused_h_args(st, src);
  if (st->tail_mode & ESPM_TAIL_RIDE) {  // the tail of the W update that produced this state rides along (include/espm_mu.h)
    ESPM_REQUIRE(st->it >= 1 && (w_update_is_local(st) || w_simplex_split(st)), "tail_mode: no local W update produced state %d", st->it);
    a.tail = make_w_tail_args(finish_args(st, 1 - src, src, st->it, 1));
    a.cs_parts = a.tail.parts;
    a.cs_nbk = a.tail.nbk;
    a.tail_on = 1;
  }
  if (a.fill_num)
    if (int rc = launch_ell_fill_num(st->gw_s, st->h[src], st->ell_fill_px, st->ell_fill_n, st->n, st->k, st->p_pad, st->log_shift,
                                     st->ell_fill_num, s))
      return rc;
  return launch_fused_ell(a, make_w_args(st), st->nblk_w, s, st->no_fused == 2, st->ell_stream);
}

int espm_mu_w_reduce(const espm_mu_state* st, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  return launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, nullptr, static_cast<hipStream_t>(stream));
}

int espm_mu_w_reduce_finalize(const espm_mu_state* st, int src, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  ESPM_REQUIRE(slot >= 0 && slot < st->hist_len, "history slot %d outside [0, %d)", slot, st->hist_len);
  const HFinalizeArgs fin = finalize_args(st, src, slot, true);
  return launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, &fin, static_cast<hipStream_t>(stream));
}

int espm_mu_w_finish(const espm_mu_state* st, int src, int hsrc, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE((src == 0 || src == 1) && (hsrc == 0 || hsrc == 1), "src/hsrc must be 0/1");
  ESPM_REQUIRE(slot < st->hist_len, "history slot %d outside [0, %d)", slot, st->hist_len);
  if (st->simplex_w && st->log_shift > 0.f) {
    const double rows = st->m > 0 ? st->m : st->n;
    if (!st->simplex_rows && rows * (double)st->log_shift >= 1.0)
      return set_error(ESPM_ENOSOLUTION, "No solution exists! (rows * log_shift >= 1)");
  }
  return launch_w_finish(finish_args(st, src, hsrc, slot, 1), static_cast<hipStream_t>(stream));
}

int espm_mu_w_reduce_finish(const espm_mu_state* st, int src, int slot, int with_finalize, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  ESPM_REQUIRE(slot >= 0 && slot + 1 < st->hist_len, "history slot %d + 1 outside [0, %d)", slot, st->hist_len);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const HFinalizeArgs fin = finalize_args(st, src, slot, true);
  WTailArgs left_out;   // ESPM_TAIL_DEFER: the caller has the tail carried by the next H-step (or espm_mu_w_update_tail)
  if (w_update_is_local(st)) {
    return launch_w_reduce_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a_slab, (size_t)st->k * st->n_pad * sizeof(float),
                                  st->nblk_w, st->a, with_finalize ? st->hpart : nullptr, nblk_h(st), with_finalize ? nullptr : st->hstat[1 - src],
                                  0, nullptr, with_finalize ? &fin : nullptr, s,   // (no riding finalize: hstat[1-src] is already reduced)
                                  (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr);
  }
  if (w_simplex_split(st))
    return w_simplex_reduce_update(st, src, slot, with_finalize ? &fin : nullptr, (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr, s);
  if (int rc = launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, with_finalize ? &fin : nullptr, s)) return rc;
  return espm_mu_w_finish(st, src, 1 - src, slot + 1, stream);
}

int espm_mu_shard_combine_finish(const espm_mu_state* st, const void* records, int world, int src, int slot,
                                 espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(records && world >= 1 && (src == 0 || src == 1), "shard_combine_finish: bad arguments");
  ESPM_REQUIRE(slot >= 0 && slot + 1 < st->hist_len, "history slot %d + 1 outside [0, %d)", slot, st->hist_len);
  WTailArgs left_out;
  if (w_update_is_local(st)) {
    return launch_w_reduce_update(finish_args(st, src, 1 - src, slot + 1, 1), records, espm_mu_shard_record_bytes(st), world, st->a,
                                  nullptr, 0, nullptr, (size_t)st->k * st->n_pad * sizeof(float), st->hstat[1 - src], nullptr,
                                  static_cast<hipStream_t>(stream), (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr);
  }
  if (w_simplex_split(st)) {   // the simplex over W with G = identity: the sum over the ranks leaves the bracket's sums, many workgroups update
    if (st->log_shift > 0.f && (double)st->n * (double)st->log_shift >= 1.0)
      return set_error(ESPM_ENOSOLUTION, "No solution exists! (rows * log_shift >= 1)");
    double* bparts = simplex_bparts(st);
    if (int rc = launch_shard_combine(records, world, espm_mu_shard_record_bytes(st), st->k * st->n_pad, st->a, st->hstat[1 - src],
                                      static_cast<hipStream_t>(stream), st->w[src], bparts, st->n, st->k, st->n_pad))
      return rc;
    return launch_w_simplex_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a, bparts, (double)ESPM_W_DICOTOMY_TOL,
                                   static_cast<hipStream_t>(stream), (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr);
  }
  if (int rc = espm_mu_shard_combine(st, records, world, 1 - src, stream)) return rc;
  return espm_mu_w_finish(st, src, 1 - src, slot + 1, stream);
}

int espm_mu_w_update_is_local(const espm_mu_state* st) {
  if (check_state(st)) return 0;
  return w_update_is_local(st) ? 1 : 0;
}

int espm_mu_w_update_tail(const espm_mu_state* st, int src, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE((src == 0 || src == 1) && slot >= 0 && slot + 1 < st->hist_len && (w_update_is_local(st) || w_simplex_split(st)), "w_update_tail: bad arguments");
  return launch_w_update_tail(make_w_tail_args(finish_args(st, src, 1 - src, slot + 1, 1)), static_cast<hipStream_t>(stream));
}

// ev (espm_mu_iterate_timed only): 3 HIP events per iteration, recorded on the launch stream ahead of the iteration's first launch,
// between its first launch (the H update - with the W accumulation where the fused kernel applies) and what follows, and behind its last.
static int iterate_impl(espm_mu_state* st, int n_iter, int final_loss, espm_stream_t stream, hipEvent_t* ev);

int espm_mu_iterate(espm_mu_state* st, int n_iter, int final_loss, espm_stream_t stream) { return iterate_impl(st, n_iter, final_loss, stream, nullptr); }

int espm_mu_iterate_timed(espm_mu_state* st, int n_iter, float* first_ms, float* rest_ms, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(n_iter >= 1 && n_iter <= 4096 && first_ms && rest_ms, "iterate_timed: 1..4096 iterations, two host arrays of that length");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev((size_t)3 * n_iter, nullptr);
  int rc = ESPM_OK;
  for (size_t i = 0; i < ev.size() && !rc; ++i) rc = check_hip(hipEventCreate(&ev[i]), "iterate_timed: event");
  if (!rc) rc = iterate_impl(st, n_iter, 0, stream, ev.data());
  if (!rc) rc = check_hip(hipStreamSynchronize(s), "iterate_timed: synchronize");
  for (int i = 0; i < n_iter && !rc; ++i) {
    rc = check_hip(hipEventElapsedTime(first_ms + i, ev[3 * i], ev[3 * i + 1]), "iterate_timed: elapsed");
    if (!rc) rc = check_hip(hipEventElapsedTime(rest_ms + i, ev[3 * i + 1], ev[3 * i + 2]), "iterate_timed: elapsed");
  }
  for (hipEvent_t e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

static int iterate_impl(espm_mu_state* st, int n_iter, int final_loss, espm_stream_t stream, hipEvent_t* ev) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(n_iter >= 0, "n_iter must be >= 0");
  ESPM_REQUIRE(st->it + n_iter < st->hist_len, "history too short: it=%d + %d >= %d", st->it, n_iter, st->hist_len);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Sparse store with a local W update: the tail of the W update (column sums of G W', rel_W: one small workgroup, 8 us as
  // a launch of its own) rides in the NEXT H-step's launch as an extra workgroup; that H-step sums the partial column sums
  // itself, and the slab reduction that follows finds colsum_gw written.  The last tail is a launch of its own.
  const bool split = w_simplex_split(st);   // (the simplex over W with G = identity: two launches after the accumulation, same tail)
  const bool defer = st->x_dtype == ESPM_X_ELL && (w_update_is_local(st) || split) && !(st->pg_q && st->pg_gamma_w > 0.f);
  // Both half-steps in one launch where the fused kernel applies (mu_fused_kernel.hpp): 2 launches per iteration instead of 3.
  const bool fused = fused_ok(st);
  bool pending = false;
  WTailArgs tail;
  for (int i = 0; i < n_iter; ++i) {
    const int cur = st->cur, slot = st->it;
    int rc;
    if (defer) {
      HStepArgs a = fused ? fused_h_args(st, cur) : make_h_args(st, cur, 1);
      if (a.fill_num && (rc = launch_ell_fill_num(st->gw_s, st->h[cur], st->ell_fill_px, st->ell_fill_n, st->n, st->k, st->p_pad,
                                                  st->log_shift, st->ell_fill_num, s)))
        return rc;
      if (pending) {
        a.cs_parts = tail.parts;
        a.cs_nbk = tail.nbk;
        a.tail_on = 1;
        a.tail = tail;
      }
      if (ev && (rc = check_hip(hipEventRecord(ev[3 * i], s), "iterate_timed: record"))) return rc;
      if (fused) {
        if ((rc = launch_fused_ell(a, make_w_args(st), st->nblk_w, s, st->no_fused == 2, st->ell_stream))) return rc;
        if (ev && (rc = check_hip(hipEventRecord(ev[3 * i + 1], s), "iterate_timed: record"))) return rc;
      } else {
        if ((rc = launch_h_ell(a, nblk_h(st), s))) return rc;
        if (ev && (rc = check_hip(hipEventRecord(ev[3 * i + 1], s), "iterate_timed: record"))) return rc;
        if ((rc = espm_mu_w_accum(st, stream))) return rc;
      }
      const HFinalizeArgs fin = finalize_args(st, cur, slot, true);
      if (split) {
        if ((rc = w_simplex_reduce_update(st, cur, slot, &fin, &tail, s))) return rc;
      } else if ((rc = launch_w_reduce_update(finish_args(st, cur, 1 - cur, slot + 1, 1), st->a_slab, (size_t)st->k * st->n_pad * sizeof(float),
                                              st->nblk_w, st->a, st->hpart, nblk_h(st), nullptr, 0, nullptr, &fin, s, &tail))) {
        return rc;
      }
      if (ev && (rc = check_hip(hipEventRecord(ev[3 * i + 2], s), "iterate_timed: record"))) return rc;
      pending = true;
    } else {
      if (ev && (rc = check_hip(hipEventRecord(ev[3 * i], s), "iterate_timed: record"))) return rc;
      if ((rc = espm_mu_step_hw(st, cur, stream))) return rc;   // (one launch where the fused kernel applies, else H-step + W accumulation)
      if (ev && (rc = check_hip(hipEventRecord(ev[3 * i + 1], s), "iterate_timed: record"))) return rc;
      // slab reduction with the H-step's finalize riding in the same launch, then (or, when W' is local, in it) the W update
      if ((rc = espm_mu_w_reduce_finish(st, cur, slot, 1, stream))) return rc;
      if (ev && (rc = check_hip(hipEventRecord(ev[3 * i + 2], s), "iterate_timed: record"))) return rc;
    }
    st->cur = 1 - cur;
    st->it = slot + 1;
  }
  if (pending)
    if (int rc = launch_w_update_tail(tail, s)) return rc;
  if (final_loss) return espm_mu_loss_only(st, st->cur, st->it, stream);
  return ESPM_OK;
}

int espm_mu_shard_exchange_finish(const espm_mu_state* st, espm_xchg* x, uint32_t seq, int src, int slot, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(x && (src == 0 || src == 1), "shard_exchange_finish: bad arguments");
  ESPM_REQUIRE(slot >= 0 && slot + 1 < st->hist_len, "history slot %d + 1 outside [0, %d)", slot, st->hist_len);
  ESPM_REQUIRE(x->record_bytes == espm_mu_shard_record_bytes(st), "shard_exchange_finish: the exchange was created for records of %zu bytes, the state packs %zu",
               x->record_bytes, espm_mu_shard_record_bytes(st));
  if (!w_update_is_local(st) && st->no_fused != 1) {
    // a dictionary G without a simplex over W: every rank contracts its own A with G and the m k results cross the links as
    // granules (w_gxchg_update_kernel): slab reduction + record reduction, exchange + W update, rows of G W' - three launches
    const WFinishArgs f = finish_args(st, src, 1 - src, slot + 1, 1);
    if (w_gsplit_applies(f)) {
      HFinalizeArgs fin = finalize_args(st, src, slot, true);
      // (this rank's statistics go to its own record in its own mailbox: the exchange launch sends them on, and writes the global ones)
      double* local = reinterpret_cast<double*>(x->mailbox + (size_t)(seq & 1u) * x->world * x->record_bytes + (size_t)x->rank * x->record_bytes +
                                                (size_t)st->k * st->n_pad * 4);
      fin.hstat_out = local;
      if (int rc = launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, &fin, static_cast<hipStream_t>(stream))) return rc;
      const int with_halo_g = st->grid_mode && st->lambda_l != 0.f;
      return launch_w_gxchg_update(f, x, seq, local, st->hstat[1 - src], st->h[1 - src], st->nx, st->ny, st->p_pad, with_halo_g,
                                   static_cast<hipStream_t>(stream));
    }
  }
  if (w_simplex_split(st) && nblk_h(st) > 0) {
    // the simplex over W with G = identity (the estimator's default constraints): the pieces of A cross the links as granules like
    // the local update's, the workgroup that sums 32 entries leaves the bracket's partials, w_simplex_update_kernel follows - two launches
    if (st->log_shift > 0.f && (double)st->n * (double)st->log_shift >= 1.0)
      return set_error(ESPM_ENOSOLUTION, "No solution exists! (rows * log_shift >= 1)");
    const HFinalizeArgs fin = finalize_args(st, src, slot, true);
    WTailArgs left_out;
    const int with_halo = st->grid_mode && st->lambda_l != 0.f;
    double* bparts = simplex_bparts(st);
    if (int rc = launch_w_exchange_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a_slab, (size_t)st->k * st->n_pad * sizeof(float), st->nblk_w,
                                          st->a, st->hstat[1 - src], fin, x, seq, st->h[1 - src], st->nx, st->ny, st->p_pad, with_halo,
                                          static_cast<hipStream_t>(stream), nullptr, bparts))
      return rc;
    return launch_w_simplex_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a, bparts, (double)ESPM_W_DICOTOMY_TOL,
                                   static_cast<hipStream_t>(stream), (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr);
  }
  if (!w_update_is_local(st) || st->no_fused == 1) {   // W' needs a global finish (G given, simplex over W): the four steps
    if (int rc = espm_mu_w_reduce_pack(st, src, slot, x->staging, stream)) return rc;
    if (int rc = espm_xchg_post(x, seq, stream)) return rc;
    if (int rc = espm_xchg_wait(x, seq, stream)) return rc;
    return espm_mu_shard_combine_finish(st, espm_xchg_records(x, (int)(seq & 1u)), x->world, src, slot, stream);
  }
  const HFinalizeArgs fin = finalize_args(st, src, slot, true);
  WTailArgs left_out;
  const int with_halo = st->grid_mode && st->lambda_l != 0.f;
  return launch_w_exchange_update(finish_args(st, src, 1 - src, slot + 1, 1), st->a_slab, (size_t)st->k * st->n_pad * sizeof(float), st->nblk_w,
                                  st->a, st->hstat[1 - src], fin, x, seq, st->h[1 - src], st->nx, st->ny, st->p_pad, with_halo,
                                  static_cast<hipStream_t>(stream), (st->tail_mode & ESPM_TAIL_DEFER) ? &left_out : nullptr);
}

int espm_mu_iterate_sharded(espm_mu_state* st, espm_xchg* x, uint32_t* seq, int n_iter, int final_loss, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(x && seq && n_iter >= 0, "iterate_sharded: bad arguments");
  ESPM_REQUIRE(st->it + n_iter < st->hist_len, "history too short: it=%d + %d >= %d", st->it, n_iter, st->hist_len);
  ESPM_REQUIRE(x->record_bytes == espm_mu_shard_record_bytes(st), "iterate_sharded: the exchange was created for records of %zu bytes, the state packs %zu",
               x->record_bytes, espm_mu_shard_record_bytes(st));
  const bool defer = st->x_dtype == ESPM_X_ELL && (w_update_is_local(st) || w_simplex_split(st)) && !(st->pg_q && st->pg_gamma_w > 0.f);
  const bool with_halo = st->grid_mode && st->lambda_l != 0.f;
  // byte offsets inside a record: [A | statistics | first owned image row | last owned image row]
  const size_t off_top = (size_t)st->k * st->n_pad * 4 + ESPM_HS_STRIDE * 8, off_bot = off_top + (size_t)st->k * (st->ny > 0 ? st->ny : 0) * 4;
  bool pending = (st->tail_mode & ESPM_TAIL_RIDE) != 0;   // the caller may hand over a deferred tail of its own last update
  int rc = ESPM_OK;
  for (int i = 0; i < n_iter && !rc; ++i) {
    const int cur = st->cur, slot = st->it;
    st->tail_mode = pending ? ESPM_TAIL_RIDE : 0;
    rc = espm_mu_step_hw(st, cur, stream);
    st->tail_mode = 0;
    const uint32_t s = ++*seq;
    const unsigned char* recs = static_cast<const unsigned char*>(espm_xchg_records(x, (int)(s & 1u)));
    st->tail_mode = defer ? ESPM_TAIL_DEFER : 0;
    // slab reduction, exchange of the records and W update: one launch when W' needs nothing but the global row sums
    if (!rc) rc = espm_mu_shard_exchange_finish(st, x, s, cur, slot, stream);
    st->tail_mode = 0;
    pending = defer;
    if (with_halo) {   // the row above this block is the LAST owned row of rank - 1, the row below the FIRST of rank + 1
      st->halo_top = x->rank > 0 ? reinterpret_cast<const float*>(recs + (size_t)(x->rank - 1) * x->record_bytes + off_bot) : nullptr;
      st->halo_bot = x->rank < x->world - 1 ? reinterpret_cast<const float*>(recs + (size_t)(x->rank + 1) * x->record_bytes + off_top) : nullptr;
    }
    st->cur = 1 - cur;
    st->it = slot + 1;
  }
  if (rc) return rc;
  if (pending)
    if ((rc = espm_mu_w_update_tail(st, 1 - st->cur, st->it - 1, stream))) return rc;
  if (final_loss) return espm_mu_loss_only(st, st->cur, st->it, stream);
  return ESPM_OK;
}

int espm_dichotomy_simplex(const double* num, const double* den, int k, int p, int den_cols, double log_shift,
                           double tol, int maxit, double* nu_out, int32_t* status_out, espm_stream_t stream) {
  ESPM_REQUIRE(num && den && nu_out && status_out, "dichotomy: NULL pointer");
  ESPM_REQUIRE(k >= 1 && p >= 1 && (den_cols == 1 || den_cols == p), "dichotomy: bad shape k=%d p=%d den_cols=%d", k,
               p, den_cols);
  if (log_shift > 0 && (double)k * log_shift >= 1.0) return set_error(ESPM_ENOSOLUTION, "No solution exists!");
  return launch_dichotomy(num, den, k, p, den_cols, log_shift, tol, maxit, nu_out, status_out,
                          static_cast<hipStream_t>(stream));
}

int espm_simplex_root_f32(const float* num, const float* den, int k, int p, float log_shift, float tol, int maxit, int fast_exit,
                          float* delta_out, float* e_out, int32_t* status_out, espm_stream_t stream) {
  ESPM_REQUIRE(num && den && delta_out && e_out && status_out, "simplex_root_f32: NULL pointer");
  ESPM_REQUIRE(k >= ESPM_MIN_K && k <= ESPM_MAX_K && p >= 1, "simplex_root_f32: bad shape k=%d p=%d (this library: %d..%d components)", k, p, ESPM_MIN_K, ESPM_MAX_K);
  if (log_shift > 0 && (double)k * log_shift >= 1.0) return set_error(ESPM_ENOSOLUTION, "No solution exists!");
  return launch_simplex_root_f32(num, den, k, p, log_shift, tol, maxit, fast_exit, delta_out, e_out, status_out, static_cast<hipStream_t>(stream));
}

int espm_surrogate_terms(const float* h_old, const float* h_new, int k, int p, int64_t ld, int nx, int ny, int grid_mode, double* part,
                         int part_doubles, double* out, espm_stream_t stream) {
  ESPM_REQUIRE(h_old && h_new && part && out && k >= 1 && k <= ESPM_KP && p >= 1 && ld >= p, "surrogate_terms: bad arguments");
  ESPM_REQUIRE(!grid_mode || (nx >= 1 && ny >= 1 && (int64_t)nx * ny == p), "surrogate_terms: grid %d x %d does not match p=%d", nx, ny, p);
  ESPM_REQUIRE(part_doubles >= (4 + ESPM_KP) * ((p + 511) / 512), "surrogate_terms: scratch of %d doubles is too small", part_doubles);
  return launch_linesearch_terms(h_old, h_new, k, p, (int)ld, nx, ny, grid_mode, nullptr, nullptr, nullptr, nullptr, part, out,
                                 static_cast<hipStream_t>(stream));
}

int espm_dichotomy_simplex_acc(double a, const double* b, const double* minus_c, int k, int p, int b_cols, double log_shift,
                               double tol, int maxit, double* nu_out, int32_t* status_out, espm_stream_t stream) {
  ESPM_REQUIRE(b && minus_c && nu_out && status_out, "dichotomy_acc: NULL pointer");
  ESPM_REQUIRE(k >= 1 && p >= 1 && (b_cols == 1 || b_cols == p) && a >= 0, "dichotomy_acc: bad arguments k=%d p=%d b_cols=%d", k, p, b_cols);
  if (log_shift > 0 && (double)k * log_shift >= 1.0) return set_error(ESPM_ENOSOLUTION, "No solution exists!");
  return launch_dichotomy_acc(a, b, minus_c, k, p, b_cols, log_shift, tol, maxit, nu_out, status_out, static_cast<hipStream_t>(stream));
}

int espm_dichotomy_simplex_pg(const double* a, int k, int p, double log_shift, double tol, int maxit, double* nu_out,
                              espm_stream_t stream) {
  ESPM_REQUIRE(a && nu_out && k >= 1 && p >= 1, "dichotomy_pg: bad arguments");
  if (log_shift > 0 && (double)k * log_shift >= 1.0) return set_error(ESPM_ENOSOLUTION, "No solution exists!");
  return launch_dichotomy_pg(a, k, p, log_shift, tol, maxit, nu_out, static_cast<hipStream_t>(stream));
}

int espm_mu_laplacian(const float* h, int k, int nx, int ny, int64_t ld, float* out, espm_stream_t stream) {
  ESPM_REQUIRE(h && out && k >= 1 && nx >= 1 && ny >= 1 && ld >= (int64_t)nx * ny, "laplacian: bad arguments");
  return launch_laplacian(h, k, nx, ny, ld, out, static_cast<hipStream_t>(stream));
}


// ---- Frobenius ("l2") branch of the step functions (mu_l2.hip) ----------------------------------------------------
static int check_l2(const espm_mu_state* st, const float* work, const double* scratch, int scratch_doubles) {
  ESPM_REQUIRE(work && scratch && scratch_doubles >= ESPM_KP * ESPM_KP, "l2 step: work (2, KP, KP) and a scratch of >= %d doubles are needed", ESPM_KP * ESPM_KP);
  ESPM_REQUIRE(st->x_dtype == ESPM_X_F32, "l2 step: the f32 store is required");
  ESPM_REQUIRE(st->xscale == 1.f, "l2 step: xscale = 1 is required");
  return ESPM_OK;
}

int espm_mu_l2_step_h(const espm_mu_state* st, int src, float* work, double* scratch, int scratch_doubles, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  if (int rc = check_l2(st, work, scratch, scratch_doubles)) return rc;
  ESPM_REQUIRE(src == 0 || src == 1, "src must be 0/1");
  ESPM_REQUIRE(st->lambda_l == 0.f && st->mu == nullptr, "l2 H step: lambda_L = 0 and mu = 0 are required (updates.py:110-114)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = launch_gram(st->gw_s, st->n, st->k, scratch, scratch_doubles, work, s)) return rc;   // GW^T GW
  HStepArgs a = make_h_args(st, src, 1);
  a.compute_loss = 0;
  a.l2_m = work;
  return dispatch_h_step(a, st->x_dtype, st->tile_px, nblk_h(st), s);
}

int espm_mu_l2_w_partials(const espm_mu_state* st, float* work, double* scratch, int scratch_doubles, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  if (int rc = check_l2(st, work, scratch, scratch_doubles)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* hh = work + ESPM_KP * ESPM_KP;
  if (int rc = launch_gram(st->h_t, st->p, st->k, scratch, scratch_doubles, hh, s)) return rc;      // H H^T
  WAccumArgs wa = make_w_args(st);
  wa.l2 = 1;
  if (int rc = dispatch_w_accum(wa, st->k, st->x_dtype, st->nblk_w, s)) return rc;                 // slabs of X H^T
  return launch_w_reduce(st->a_slab, st->a, st->nblk_w, st->k * st->n_pad, nullptr, s);
}

int espm_mu_l2_w_finish(const espm_mu_state* st, int src, const float* gtg, const float* work, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(work && (src == 0 || src == 1), "l2 W finish: bad arguments");
  ESPM_REQUIRE(st->m == 0 || gtg, "l2 W step: G^T G (m, m) is needed when G is given");
  ESPM_REQUIRE(!st->simplex_w, "l2 W step: no simplex over W in the Frobenius branch (updates.py:31-36)");
  return launch_w_finish_l2(st->a, st->n, st->n_pad, st->m, st->k, st->m > 0 ? st->g : nullptr, gtg, work + ESPM_KP * ESPM_KP, st->w[src],
                            st->w[1 - src], st->fixed_w, st->log_shift, static_cast<hipStream_t>(stream));
}

int espm_mu_l2_step_w(const espm_mu_state* st, int src, const float* gtg, float* work, double* scratch, int scratch_doubles,
                      espm_stream_t stream) {
  if (int rc = espm_mu_l2_w_partials(st, work, scratch, scratch_doubles, stream)) return rc;
  return espm_mu_l2_w_finish(st, src, gtg, work, stream);
}

int espm_mu_linesearch_terms(const espm_mu_state* st, int hold, int hnew, double* out, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(out && (hold == 0 || hold == 1) && hnew == 1 - hold, "linesearch_terms: bad arguments");
  ESPM_REQUIRE(!st->halo_top && !st->halo_bot && (st->p_total == 0 || st->p_total == st->p), "linesearch_terms: not built for a sharded image");
  // the H-step's record buffer is free between espm_mu_h_finalize and the next espm_mu_step_h: ESPM_HP_STRIDE rows of
  // ceil(p / tile_px) >= ceil(p / 512) doubles hold the 3 + KP rows of partials
  return launch_linesearch_terms(st->h[hold], st->h[hnew], st->k, st->p, st->p_pad, st->nx, st->ny, st->grid_mode, nullptr, nullptr,
                                 nullptr, nullptr, st->hpart, out, static_cast<hipStream_t>(stream));
}

int espm_mu_linesearch_terms_sharded(const espm_mu_state* st, int hold, int hnew, const float* old_halo_top, const float* old_halo_bot,
                                     double* out, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(out && (hold == 0 || hold == 1) && hnew == 1 - hold, "linesearch_terms_sharded: bad arguments");
  // the boundary rows travel in the records only when the Laplacian term is on (espm_mu_shard_record_bytes)
  ESPM_REQUIRE(!st->grid_mode || st->lambda_l != 0.f, "linesearch_terms_sharded: lambda_L = 0 leaves the boundary rows out of the records");
  ESPM_REQUIRE((old_halo_top == nullptr) == (st->halo_top == nullptr) && (old_halo_bot == nullptr) == (st->halo_bot == nullptr),
               "linesearch_terms_sharded: the old H's boundary rows must be given where the new H has them");
  return launch_linesearch_terms(st->h[hold], st->h[hnew], st->k, st->p, st->p_pad, st->nx, st->ny, st->grid_mode, old_halo_top,
                                 old_halo_bot, st->halo_top, st->halo_bot, st->hpart, out, static_cast<hipStream_t>(stream));
}

size_t espm_mu_shard_record_bytes(const espm_mu_state* st) {
  if (check_abi(st)) return 0;
  size_t b = (size_t)st->k * st->n_pad * 4 + ESPM_HS_STRIDE * 8 + 2 * (size_t)st->k * (st->ny > 0 ? st->ny : 0) * 4;
  return (b + 15) / 16 * 16;
}

int espm_mu_shard_pack(const espm_mu_state* st, int hnew, void* record, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(record && (hnew == 0 || hnew == 1), "shard_pack: bad arguments");
  const int with_halo = st->grid_mode && st->lambda_l != 0.f;
  return launch_shard_pack(st->a, st->hstat[hnew], st->h[hnew], st->k, st->n_pad, st->nx, st->ny, st->p_pad,
                           with_halo, record, static_cast<hipStream_t>(stream));
}

int espm_mu_w_reduce_pack(const espm_mu_state* st, int src, int slot, void* record, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(record && (src == 0 || src == 1), "w_reduce_pack: bad arguments");
  ESPM_REQUIRE(slot >= 0 && slot < st->hist_len, "history slot %d outside [0, %d)", slot, st->hist_len);
  HFinalizeArgs fin = finalize_args(st, src, slot, true);
  fin.hstat_out = reinterpret_cast<double*>(static_cast<unsigned char*>(record) + (size_t)st->k * st->n_pad * 4);
  const int with_halo = st->grid_mode && st->lambda_l != 0.f;
  return launch_w_reduce_pack(st->a_slab, st->nblk_w, st->k, st->n_pad, fin, st->h[1 - src], st->nx, st->ny, st->p_pad, with_halo,
                              record, static_cast<hipStream_t>(stream));
}

int espm_mu_shard_combine(const espm_mu_state* st, const void* records, int world, int hnew, espm_stream_t stream) {
  if (int rc = check_state(st)) return rc;
  ESPM_REQUIRE(records && world >= 1 && (hnew == 0 || hnew == 1), "shard_combine: bad arguments");
  return launch_shard_combine(records, world, espm_mu_shard_record_bytes(st), st->k * st->n_pad, st->a,
                              st->hstat[hnew], static_cast<hipStream_t>(stream));
}

}  // extern "C"
