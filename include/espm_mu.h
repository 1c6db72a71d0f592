/*
 * espm_mu.h - C ABI of libespm_mu.so: the SmoothNMF multiplicative-update hot path of
 * adriente/espm, written for AMD Instinct MI355X (gfx950 / CDNA4).
 *
 * The reference (pure Python/numpy, no FFI of its own) has exactly one compute path worth
 * a native boundary; the entry points below are what a binding for that path replaces.
 * Citations are file:line in the reference repository (adriente/espm @ v1.1.3):
 *
 *   espm_mu_step_h        <- espm/estimators/updates.py:83-156  multiplicative_step_h (KL branch)
 *                            + espm/estimators/dicotomy.py:4-55,111-173 (per-pixel simplex root)
 *                            + espm/utils.py:39-76 (Laplacian, as a stencil)
 *                            + espm/measures.py:456-504,524-548,560-577 (loss pieces of the INPUT state)
 *   espm_mu_h_finalize    <- espm/estimators/base.py:167-207, smooth_nmf.py:457-475 (loss assembly)
 *   espm_mu_w_accum       <- espm/estimators/updates.py:38-39,53-59 (R = X/(GWH), R H^T)
 *   espm_mu_w_reduce      <- (new) fixed-order reduction of the per-workgroup partial R H^T slabs
 *   espm_mu_w_finish      <- espm/estimators/updates.py:58-76 (G^T., simplex_W, clamp, fixed_W)
 *                            + updates.py:38 / base.py:189 (GW = G @ W) + base.py:323 (rel_W)
 *   (rel_H, base.py:324, is produced by espm_mu_step_h / espm_mu_h_finalize of the NEXT state evaluation)
 *   espm_mu_hstat         <- updates.py:139 (max_j H), updates.py:60 (sum_j H)
 *   espm_mu_step_hw       <- smooth_nmf.py:324-339 + :404-414 up to the sum over pixels: the H update and, with the new H, the
 *                            pixel contraction of the W update (updates.py:38-39, :53-59) in one launch
 *   espm_mu_iterate       <- espm/estimators/smooth_nmf.py:284-455 (_iteration, log_surrogate)
 *                            driven by base.py:313-394 (single GPU, no host sync)
 *   espm_mu_shard_*, espm_xchg_*, espm_mu_iterate_sharded
 *                         <- (new) pixel-row sharding over the GPUs of a node and its record exchange; no reference analogue
 *   espm_dichotomy_simplex<- espm/estimators/dicotomy.py:4-55 (module-level function)
 *   espm_simplex_root_f32 <- dicotomy.py:4-55 as the H update solves it: per pixel, fp32, in the shifted unknown (a launch of its own for tests)
 *   espm_mu_pack_x        <- base.py:243-247 (validate_data / hspy_comp transpose) as a layout step
 *   espm_mu_laplacian     <- espm/utils.py:39-76 applied to H (H @ L), measures.py:560-577
 *   espm_mu_w_reduce_finish, espm_mu_shard_combine_finish, espm_mu_w_reduce_pack
 *                         <- (new) the W-step after the accumulation in one call / launch (updates.py:58-76 folded into
 *                            the slab or rank-record reduction when W' needs nothing global)
 *   espm_mu_linesearch_terms, espm_mu_linesearch_terms_sharded
 *                         <- espm/estimators/surrogates.py:6-149 + smooth_nmf.py:376-381 (linesearch=True; a rank's rows)
 *   espm_mu_l2_step_h / _w, espm_mu_l2_w_partials / _finish
 *                         <- espm/estimators/updates.py:109-118, :31-36 (Frobenius branch, l2=True; the W step in two halves
 *                            around the sum over the ranks of a sharded image)
 *   state fields breg_sr_* <- updates.py:40-48, :120-125 (Bregman variant, algo = "bmd")
 *   state field h_rule = 1 <- updates.py:263-315 + dicotomy.py:57-82 (multiplicative_step_hq, algo = "l2_surrogate")
 *   h_rule = 2, pg_gamma_w, pg_q <- updates.py:317-395 + dicotomy.py:84-108 (proj_grad_step_h / _w, algo = "projected_gradient")
 *                            + smooth_nmf.py:382-401, :438-447 (its linesearch)
 *   espm_dichotomy_simplex_acc / _pg <- dicotomy.py:57-108 (module-level functions)
 *   espm_surrogate_terms  <- espm/estimators/surrogates.py:6-149 (module-level surrogates)
 *   espm_lu_pl            <- espm/estimators/updates.py:179 -> scikit-learn's _initialize_nmf -> _randomized_range_finder: the LU
 *                            normaliser of its power iterations (scipy.linalg.lu(A, permute_l=True)[0]) on tall device matrices
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  All array pointers are DEVICE pointers owned by the
 *     caller (e.g. torch tensors); the library allocates nothing that outlives a call.
 *   - Every call only enqueues work on `stream` (a hipStream_t) and returns; no host sync.
 *   - Return value: 0 on success, negative espm_status otherwise; espm_mu_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - X is kept twice in HBM, once in the order each half-step streams it.  Dense stores: channel-major inside
 *     pixel blocks (p_pad / x_tile, n_cm, x_tile) for the H-step and pixel-major (p, n_pad) for the W-step, as u8
 *     (integer counts <= 255), bf16 or fp32, whichever is lossless.  Sparse count store (ESPM_X_ELL): 16-bit lists
 *     of the non-zero entries by pixel and by (pixel block, channel), described at the state fields below.
 *   - W, H, G, GW and every accumulator are fp32; loss sums are fp64.
 */
#ifndef ESPM_MU_H
#define ESPM_MU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* espm_stream_t; /* hipStream_t */

typedef enum espm_status {
  ESPM_OK = 0,
  ESPM_EINVAL = -1,       /* bad argument / shape (reference: ValueError / AssertionError) */
  ESPM_ENOSOLUTION = -2,  /* k * log_shift >= 1 (dicotomy.py:22-23 "No solution exists!") */
  ESPM_EHIP = -3,         /* HIP runtime error */
  ESPM_EUNSUPPORTED = -4  /* configuration not built (k > ESPM_MAX_K) */
} espm_status;

enum { ESPM_X_F32 = 0, ESPM_X_BF16 = 1, ESPM_X_U8 = 2 /* integer counts <= 255 */,
       ESPM_X_ELL = 3 /* integer counts, non-zero entries only (sparse count store, below) */ };
enum { ESPM_SRC_F32 = 0, ESPM_SRC_F64 = 1 };
enum { ESPM_LAYOUT_CM = 0 /* (n, p) channel-major */, ESPM_LAYOUT_PM = 1 /* (p, n) pixel-major */ };

/* The library is built three times from the same sources: libespm_mu.so (1..8 components, the strides below = 8),
 * libespm_mu_wide32.so (-DESPM_KP=32 -DESPM_MIN_K=17 -DESPM_MAX_K=32: 17..32 components; the dense stores only, H tiles of 128 pixels, on the
 * 8-bit and bf16 stores the matrix-core kernels only - espm/estimators/base.py:126-132 has no limit on n_components) and
 * libespm_mu_wide.so (-DESPM_KP=16 -DESPM_MIN_K=9 -DESPM_MAX_K=16: 9..16 components; the sparse store's LDS table then has
 * rows of 12 or 16 floats and must fit ESPM_ELL_LDS_MAX together with the numerators of a tile).  Same entry points, same state struct; every
 * size below that names KP follows the build. */
#ifndef ESPM_KP
#define ESPM_KP 8          /* padded component stride of GW (n_pad, KP) and H^T (p, KP): 8, 16 or 32 */
#endif
#ifndef ESPM_MAX_K
#define ESPM_MAX_K ESPM_KP /* components supported by the built kernels */
#endif
#ifndef ESPM_MIN_K
#define ESPM_MIN_K 1
#endif
#define ESPM_PPAD 512      /* p_pad is a multiple of this */
#define ESPM_NPAD 8        /* n_pad is a multiple of this */
#define ESPM_ELL_TILE 512  /* sparse store: pixels per H-step workgroup (8 lists of 64 pixels)               */
#define ESPM_ELL_PB 1024   /* sparse store at its full geometry: pixels per block of the W accumulation (state field ell_pb) */
#define ESPM_ELL_PBITS 10  /* log2(ESPM_ELL_PB)                                                             */
#define ESPM_ELL_UNIT_ROWS 8 /* sparse store: the unit rows of a list group are a multiple of this (16 entries: one per bank quad) */
#define ESPM_ELL_UNIT_MAX_N 4080 /* sparse store: H-step lists have unit rows when n <= this (index << 4 < 2^16, and at most 255 ones in any of the builder's 16 residue classes of a list: its 8-bit counters) */
#define ESPM_ELL_PAIR_MAX_K 6 /* sparse store H-step: list groups are walked in pairs (2 partial numerators) up to this k */
#define ESPM_ELL_STREAM_BYTES (256 << 20) /* sparse store: list bytes beyond which a caller sets espm_mu_state.ell_stream (the MI355X's last-level cache) */
#define ESPM_FUSED_MIN_PB 512 /* sparse store: W blocks (ell_pb) from which the fused launch is the default whatever their number           */
#define ESPM_FUSED_MIN_BLOCKS 192 /* ... and smaller blocks when there are at least this many (one per CU, or nearly: a 64-row shard of the
                                   * headline image has 256 of 128 pixels; config 2's 128 blocks leave half the chip to the two launches,
                                   * which then win: 27.6 against 35.2 us per iteration, profiles/r03e_c2_iter.log)                              */
#define ESPM_ELL_WTHREADS 1024 /* threads of a W-accumulation workgroup of the sparse store (16 waves)      */
#ifndef ESPM_ELL_BUCKETS
#define ESPM_ELL_BUCKETS 16    /* residue classes of the index by which a list's unit elements are placed in its unit rows (mu_ell_build.hip) */
#endif
#define ESPM_ELL_LDS_MAX (160 * 1024) /* LDS bytes the sparse H-step may use (GW table + numerators): a workgroup's LDS on gfx950 (144 KB until round 5) */
#define ESPM_NCM 16        /* channel rows of x_cm are padded to a multiple of this */
#define ESPM_W_DICOTOMY_TOL 1e-5f /* tolerance of the simplex multiplier of the W update: espm/conf.py dicotomy_tol, which the
                                     reference's multiplicative_step_w always uses (updates.py:61-68); the state's
                                     dicotomy_tol is the H update's (the estimator's argument, smooth_nmf.py) */
#define ESPM_TAIL_DEFER 1  /* espm_mu_state.tail_mode bits */
#define ESPM_TAIL_RIDE 2

/* per-workgroup partial record written by the H-step (doubles), stored field-major:
 * hpart[field * nblk + block] */
#define ESPM_HP_KL 0       /* sum X*log2(X/Y) over the block (input state)            */
#define ESPM_HP_REG 1      /* sum_k mu_k log(H_in + eps_reg)                           */
#define ESPM_HP_LAP 2      /* sum H_in * (H_in L)                                      */
#define ESPM_HP_BAD 3      /* count of non-finite H_out entries                        */
#define ESPM_HP_ROWSUM 4   /* [4, 4+KP): sum_j H_out[k, j]                             */
#define ESPM_HP_MAX (4 + ESPM_KP)      /* [4+KP, 4+2KP): max_j H_out[k, j]  (12 with KP = 8)           */
#define ESPM_HP_RELH (4 + 2 * ESPM_KP) /* (20) max |H_in - H_prev| / (H_in + tol mean H_in) over the block (base.py:324) */
#define ESPM_HP_PGQ (5 + 2 * ESPM_KP)  /* (21) projected-gradient rule only: sum <H' - H, grad> + gamma_H ||H' - H||^2 over the block */
#define ESPM_HP_RELW (6 + 2 * ESPM_KP)  /* (22) the block's share of rel_W of the W update that produced the input state (max over its slice of
                                         * the entries of W, base.py:323), where the launch carried that update's tail; else -1           */
#define ESPM_HP_NSCALAR 5  /* KL, REG, LAP, BAD, RELH                                  */
#define ESPM_HP_STRIDE (8 + 2 * ESPM_KP) /* (24) */

/* per-state statistics of one H buffer (doubles): produced by espm_mu_hstat / h_finalize */
#define ESPM_HS_ROWSUM 0   /* [0, KP)  */
#define ESPM_HS_MAX ESPM_KP /* [KP, 2 KP)  */
#define ESPM_HS_STRIDE (2 * ESPM_KP)

/* history record per evaluated state (doubles) */
#define ESPM_HI_KLX 0      /* xscale * sum X ln(X/Y)   (local pixels)                  */
#define ESPM_HI_REG 1      /* sum mu log(H+eps)        (local pixels)                  */
#define ESPM_HI_LAP 2      /* sum H*(HL)               (local pixels)                  */
#define ESPM_HI_SUMY 3     /* sum_k colsum(GW)_k * rowsum(H)_k  (uses the global rowsum in hstat) */
#define ESPM_HI_BAD 4      /* non-finite entries produced by the step that followed     */
#define ESPM_HI_REL_W 5    /* base.py:323 for the update that PRODUCED this state       */
#define ESPM_HI_REL_H 6    /* base.py:324 for the update that produced this state; evaluated by the H-step that
                              starts FROM this state (local max; max-reduce over ranks) */
#define ESPM_HI_STRIDE 8

/* Version of this header's binary interface: the layout of espm_mu_state and the meaning of its fields.  Every entry
 * point that takes a state checks st->struct_size == sizeof(espm_mu_state) and st->abi_version == ESPM_MU_ABI_VERSION
 * first and fails with ESPM_EINVAL otherwise: a binding whose copy of the layout has drifted is refused instead of
 * having its pointers misread.  A binding can also compare its layout field by field with espm_mu_state_layout(). */
#define ESPM_MU_ABI_VERSION 4

typedef struct espm_mu_state {
  uint32_t struct_size;   /* sizeof(espm_mu_state) as the CALLER sees it                  */
  uint32_t abi_version;   /* ESPM_MU_ABI_VERSION the caller was written against            */
  /* geometry */
  int32_t n;        /* energy channels: rows of X                                       */
  int32_t m;        /* columns of G; 0 means G = identity (then W is (n, k))            */
  int32_t k;        /* components, 1..ESPM_MAX_K                                        */
  int32_t p;        /* local pixels (= nx*ny when grid_mode = 1)                        */
  int32_t nx, ny;   /* local image rows, row length                                     */
  int32_t n_pad;    /* roundup(n, ESPM_NPAD)                                            */
  int32_t p_pad;    /* roundup(p, ESPM_PPAD)                                            */
  int32_t x_dtype;  /* ESPM_X_F32 | ESPM_X_BF16 | ESPM_X_U8 | ESPM_X_ELL                */
  int32_t tile_px;  /* H-step pixel tile per workgroup: 64 * {1,2,4,8}, see query       */
  int32_t nblk_w;   /* pixel blocks of the W accumulation (rows of a_slab)              */
  int32_t x_tile;   /* pixel-block width of the tile-major x_cm (multiple of tile_px)   */
  int32_t n_cm;     /* roundup(n, ESPM_NCM): channel rows per pixel block of x_cm       */
  int32_t h_variant; /* must be 0 (1 was the retired matrix-core H-step) */
  int64_t p_total;  /* pixels of the whole image over all ranks (= p on one GPU)        */
  /* flags */
  int32_t simplex_h, simplex_w;
  int32_t grid_mode;        /* 0: L = identity (shape_2d None, base.py:289-291), 1: 2-D stencil */
  int32_t compute_loss;     /* H-step also accumulates the KL term (1 log per element)  */
  /* scalars */
  float lambda_l, sigma_l, eps_reg, log_shift, dicotomy_tol, rel_tol;
  float xscale;     /* X_ = xscale * X_stored (normalize=True, base.py:264-267)         */
  float gw_floor;   /* lower clamp of GW entries, keeps X/(GW H) finite (updates.py:129-131) */
  /* data */
  const void* x_cm;         /* (p_pad / x_tile, n_cm, x_tile) u8|bf16|f32: channel-major inside pixel blocks, zero padded */
  const void* x_pm;         /* (p, n_pad) bf16|f32, zero padded                         */
  const float* g;           /* (n, m) row-major or NULL                                  */
  const float* colsum_g;    /* (m) or NULL                                               */
  float* w[2];              /* (m or n, k) row-major, ping-pong                          */
  float* gw_s;              /* (n_pad, KP): GW / xscale, pad rows = 1                    */
  double* colsum_gw;        /* (KP): column sums of GW over real rows                    */
  void* gw_a;               /* reserved (NULL): operands of the retired matrix-core H-step */
  float* gw_p;              /* reserved (NULL) */
  float* h[2];              /* (k, p_pad), ping-pong, pad columns must be positive       */
  float* h_t;               /* (p, KP): transposed copy of the newest H                  */
  const float* mu;          /* (k) or NULL                                               */
  const float* fixed_h;     /* (k, p_pad), entries >= 0 are imposed; or NULL             */
  const float* fixed_w;     /* (m or n, k) or NULL                                       */
  const int32_t* simplex_rows; /* (m or n) 0/1 mask of rows under simplex_W, NULL = all  */
  const float* halo_top;    /* (k, ny) image row above the local block or NULL           */
  const float* halo_bot;    /* (k, ny) image row below the local block or NULL           */
  /* workspaces */
  double* hpart;            /* (ESPM_HP_STRIDE, ceil(p / tile_px)) field-major              */
  double* hstat[2];         /* (ESPM_HS_STRIDE) statistics of h[0] / h[1] (global)       */
  float* a_slab;            /* (nblk_w, k, n_pad)                                        */
  float* a;                 /* (k, n_pad)                                                */
  float* w_scratch;         /* (2, m or n, k), zero before the first call (a ticket of the dictionary-G W finish lives there) */
  double* hist;             /* (hist_len, ESPM_HI_STRIDE), zero-initialised              */
  int32_t hist_len;
  int32_t cur;              /* index of the current W/H buffers (0/1), flipped by iterate */
  int32_t it;               /* number of completed iterations = history slot of the current state */
  /* Sparse count store (x_dtype = ESPM_X_ELL; x_cm / x_pm are then unused and may be NULL).  Only the non-zero
   * entries of X are kept, 16 bits each, as lists padded to the longest of 64 ("ELL"): the entries 2r and 2r+1
   * of the 64 lists of a wave form row r of 64 dwords (low half first), value 0 = padding.  A count that
   * exceeds its field is split over several entries with the same index (the kernels evaluate the loss term
   * x_i log2(x_i / Y) per entry; ell_klc restores sum x log2(x / Y) for split counts).
   *   H-step: one list per pixel; entry = count << ell_cbits | channel.  Inside every window of tile_px pixels
   *           (the pixels of one H-step workgroup) the lists are ordered by decreasing length: slot s of the
   *           window starting at pixel w0 is pixel w0 + pix_perm[w0 + s], and the 64 lists of a wave are 64
   *           consecutive slots, so they have about the same length (little padding).  Rows
   *           [ell_h_off[2 g], ell_h_off[2 g + 2]) belong to slots 64 g .. 64 g + 63.
   *   W-step: one list per (block of ell_pb = 2 tile_px pixels, channel); inside block b the 64 lists of a wave are
   *           the channels chan_perm[b][64 cg .. 64 cg + 63] (the block's channels by decreasing list length,
   *           -1 = none); entry = count << log2(ell_pb) | pixel - block start; rows
   *           [ell_w_off[2 i], ell_w_off[2 i + 2]) with i = b * n_cg + cg.  nblk_w = ceil(p / ell_pb).
   *   UNIT rows: most non-zero entries of a count image are ones.  The first rows [off[2 i], off[2 i + 1]) of
   *           a group hold only entries with count 1, in EVERY lane and position (no padding), stored without a
   *           count as index << 4 (the byte offset of a 16-byte table row): u = min over the 64 lists of their
   *           elements equal to 1, unit rows = ESPM_ELL_UNIT_ROWS * floor(u / (2 ESPM_ELL_UNIT_ROWS)); a list's
   *           2 * (unit rows) of a list's ones go there - which ones and in which order is the builder's choice
   *           (espm_mu_ell_fill places the ones of lane l with index = q mod 16 at positions = q - l mod 16, so
   *           that the 16 lanes of a ds_read_b128 group address 16 different bank quads of the table) - all its
   *           other elements go to the general rows [off[2 i + 1], off[2 i + 2]) in the count << bits | index
   *           form.  H-step lists have unit rows only when n <= ESPM_ELL_UNIT_MAX_N. */
  const uint32_t* ell_h;    /* (rows_h, 64) */
  const int32_t* ell_h_off; /* (2 p_pad / 64 + 1) */
  const float* ell_klc;     /* (p_pad): loss correction of split counts per pixel: sum over the elements of
                               x log2 x minus the sum over their entries x_i of x_i log2 x_i (0 without splits) */
  const uint32_t* ell_w;    /* (rows_w, 64) */
  const int32_t* ell_w_off; /* (2 nblk_w * n_cg + 1) */
  const int32_t* chan_perm; /* (nblk_w, 64 * n_cg) */
  int32_t ell_cbits;        /* index bits of an H-step entry: 2^ell_cbits >= n, <= 14 */
  int32_t n_cg;             /* channel groups: ceil(n / 64) */
  const int32_t* pix_perm;  /* (p_pad): slot -> pixel offset inside its tile_px window (H-step lists) */
  const float* g_t;         /* optional (m, n_pad): G transposed, zero padded; the W finish then reads G with
                               coalesced loads (NULL: it reads g with a stride of m) */
  /* Bregman variant of both updates (updates.py:40-48, :120-125; algo = "bmd"), G = identity only (the reference's own W
   * step needs a square G).  Sums of the STORED X, both or neither:
   *   H: num = sR / H, denum = colsum(GW) - GW^T (X / GWH) + sR / H      with sR_j = xscale * breg_sr_px[j] = sum_c X_cj
   *   W: W' = sR W / ((rowsum(H) - (X / GWH) H^T) W + sR), no simplex    with sR_c = xscale * breg_sr_ch[c] = sum_j X_cj */
  const float* breg_sr_px;  /* (p_pad) or NULL */
  const float* breg_sr_ch;  /* (n) or NULL */
  int32_t h_rule;           /* H update: 0 = log surrogate (multiplicative_step_h, updates.py:83-156), 1 = quadratic surrogate
                               of the Laplacian term (multiplicative_step_hq, updates.py:263-315: positive root of
                               a H'^2 + b H' - c = 0, its own simplex multiplier dicotomy.py:57-82; mu only enters the loss),
                               2 = projected gradient (proj_grad_step_h, updates.py:372-395: H - grad / gamma_H with
                               gamma_H in sigma_l, projection on the simplex dicotomy.py:84-108) */
  float pg_gamma_w;         /* > 0: the W update is the projected-gradient step W - grad / pg_gamma_w, clamped
                               (proj_grad_step_w, updates.py:353-370; no simplex over W); 0: multiplicative update */
  double* pg_q;             /* optional (hist_len, 2): terms of the projected gradient's linesearch (smooth_nmf.py:382-401,
                               :438-447; surrogates.py:153-170).  [t][0] = sum <H' - H, grad_H> + gamma_H ||H' - H||^2 of
                               the H update that STARTS from state t (written by espm_mu_h_finalize(.., slot = t));
                               [t][1] = the same for the W update that PRODUCED state t.  The caller adds the losses. */
  /* Sparse store, pixels without a single count.  The reference fills them with log_shift in every channel
   * (base.py:519-528); under simplex_H that fill alone decides their column of H.  The lists stay empty for such a
   * pixel; instead ell_klc[pixel] = -(1 + i) names entry i of ell_fill_px, and espm_mu_step_h first forms the fill's
   * numerator log_shift * sum_c GW_c / (GW_c . H_pixel) of the ell_fill_n listed pixels into ell_fill_num
   * (k, ell_fill_n), which the H-step's epilogue adds.  The fill's part in the W update and in the loss is O(log_shift)
   * and is left out.  ell_fill_n = 0: no such pixel, both pointers may be NULL. */
  const int32_t* ell_fill_px;
  float* ell_fill_num;
  int32_t ell_fill_n;
  /* Tail of the local W update (espm_mu_w_update_is_local: column sums of G W' -> colsum_gw, rel_W -> hist): one small
   * workgroup, 8 us as a launch of its own.  bit 0 (ESPM_TAIL_DEFER): espm_mu_w_reduce_finish / espm_mu_shard_combine_finish
   * leave it out.  bit 1 (ESPM_TAIL_RIDE): the next espm_mu_step_h / espm_mu_loss_only (sparse store) carries the tail of
   * the update that PRODUCED state `it` (w[1 - src] -> w[src]) as an extra workgroup and sums the partial column sums
   * itself; the caller sets the bit for that one call, or flushes with espm_mu_w_update_tail.  espm_mu_iterate does all
   * this internally and ignores the field. */
  int32_t tail_mode;
  /* Sparse store, default H rule: espm_mu_iterate and espm_mu_step_hw run both half-steps of an iteration in ONE launch -
   * the workgroup that has updated the ell_pb pixels of a block (two H tiles) goes straight on with that block's part of
   * R H^T, which needs no other pixel's new H (updates.py:38-39, :53-59); h_t is then not written.  no_fused = 1 keeps the
   * two launches (A/B, tests); 2 runs the fused kernel with a fixed assignment of its work units to waves instead of the
   * dynamic one (A/B only). */
  int32_t no_fused;
  /* Sparse store: pixels per block of the W accumulation = 2 tile_px (espm_mu_query): ESPM_ELL_PB = 1024 for an image that
   * fills the chip with 512-pixel H tiles, 128 .. 512 for smaller images and shards, so that there is about one block per
   * compute unit and one workgroup can own the block through both half-steps. */
  int32_t ell_pb;
  /* Sparse store: 1 = the lists (256 bytes per row of ell_h and ell_w: the rows[2] of espm_mu_ell_plan) do not fit the device's
   * last-level cache, i.e. exceed ESPM_ELL_STREAM_BYTES: the one-launch iteration then reads them with non-temporal loads, which
   * leave that cache to what IS read again (measured: -1.4 % at 2048 x 512^2, where the lists are 0.5 GB; +15 % if set on a
   * 64-row shard of it, whose lists stay in the cache from one iteration to the next).  A hint: results are the same bits
   * either way, and a launch without a streamed form (blocks below ESPM_ELL_PB pixels, the generic instances) ignores it. */
  int32_t ell_stream;
} espm_mu_state;

const char* espm_mu_version(void);
const char* espm_mu_last_error(void);
/* sizeof(espm_mu_state) and ESPM_MU_ABI_VERSION of the library, and its view of the layout as text:
 * "name:offset:size;" for every field in declaration order (arrays: the whole array), NUL-terminated, static storage. */
size_t espm_mu_state_size(void);
int espm_mu_abi_version(void);
const char* espm_mu_state_layout(void);

/* Fills n_pad, p_pad, tile_px, x_tile, nblk_w of `st` from n, p, k, x_dtype and the device's CU count. */
int espm_mu_query(espm_mu_state* st);

/* X (host layout, device memory) -> the two padded device layouts.  src is (n, p) when
 * src_layout = CM or (p, n) when PM (hyperspy's layout), leading dimension ld (elements). */
int espm_mu_pack_x(const void* src, int src_dtype, int src_layout, int64_t ld, int n, int p,
                   void* x_cm, void* x_pm, int x_dtype, int n_pad, int p_pad, int x_tile, int n_cm,
                   espm_stream_t stream);

/* Sparse count store (x_dtype = ESPM_X_ELL) from the dense pixel-major 8-bit matrix x_pm_u8 (p, n_pad) that
 * espm_mu_pack_x writes for x_dtype = ESPM_X_U8 (its x_cm argument may be NULL then).  `st` needs n, p, x_dtype =
 * ESPM_X_ELL and espm_mu_query.  Three steps, all buffers caller-allocated:
 *   count: cnt_px (2, p_pad) entries of each pixel's H-step list, then its elements equal to 1; cnt_bc (2, nblk_w,
 *          64 n_cg) the same for each (pixel block, channel) W-step list (natural channel order); ell_klc (p_pad).
 *   plan : chan_perm (nblk_w, 64 n_cg), pix_perm (p_pad; windows of st->tile_px pixels), ell_h_off (2 p_pad / 64 + 1),
 *          ell_w_off (2 nblk_w n_cg + 1) and rows[2] (device):
 *          rows of 64 dwords of the H-step and of the W-step lists.  The caller reads rows[] back, checks
 *          64 rows < 2^31 and allocates ell_h (rows[0], 64) and ell_w (rows[1], 64), ZERO-initialised.
 *   fill : writes the entries.  The dense x_pm_u8 can be released afterwards.  Optional, for speed: st->x_cm != NULL hands the fill the
 *          same 8-bit counts channel-major, [p_pad / ESPM_PPAD][n_cm][ESPM_PPAD] - espm_mu_pack_x's x_cm output with x_tile = ESPM_PPAD -
 *          from which the channel lists are read with 16-byte loads (the lists come out the same); reset st->x_cm to NULL afterwards. */
int espm_mu_ell_count(const espm_mu_state* st, const void* x_pm_u8, int32_t* cnt_px, int32_t* cnt_bc, float* ell_klc,
                      espm_stream_t stream);
/* The same two steps with the lists' histograms of unit elements handed from count to fill, which saves the fill its first of two passes
 * over X: ESPM_ELL_BUCKETS bytes per list - its elements equal to 1 per residue class of their index (channel; pixel inside the block) mod
 * ESPM_ELL_BUCKETS, what the placement of the unit rows goes by.  bkt_px (p_pad, ESPM_ELL_BUCKETS) and bkt_bc (nblk_w, 64 n_cg,
 * ESPM_ELL_BUCKETS) bytes, caller-allocated, both or neither (NULL, NULL = the functions without _hist).  The lists come out the same. */
int espm_mu_ell_count_hist(const espm_mu_state* st, const void* x_pm_u8, int32_t* cnt_px, int32_t* cnt_bc, float* ell_klc, uint8_t* bkt_px,
                           uint8_t* bkt_bc, espm_stream_t stream);
int espm_mu_ell_fill_hist(const espm_mu_state* st, const void* x_pm_u8, const int32_t* chan_perm, const int32_t* pix_perm,
                          const int32_t* ell_h_off, const int32_t* ell_w_off, uint32_t* ell_h, uint32_t* ell_w, const uint8_t* bkt_px,
                          const uint8_t* bkt_bc, espm_stream_t stream);
int espm_mu_ell_plan(const espm_mu_state* st, const int32_t* cnt_px, const int32_t* cnt_bc, int32_t* chan_perm,
                     int32_t* pix_perm, int32_t* ell_h_off, int32_t* ell_w_off, int64_t* rows, espm_stream_t stream);
int espm_mu_ell_fill(const espm_mu_state* st, const void* x_pm_u8, const int32_t* chan_perm, const int32_t* pix_perm,
                     const int32_t* ell_h_off, const int32_t* ell_w_off, uint32_t* ell_h, uint32_t* ell_w,
                     espm_stream_t stream);

/* statistics (row sums, row maxima) of st->h[which] into st->hstat[which] (local pixels). */
int espm_mu_hstat(const espm_mu_state* st, int which, espm_stream_t stream);

/* gw_s, colsum_gw from g, w[which] (updates.py:107). */
int espm_mu_build_gw(const espm_mu_state* st, int which, espm_stream_t stream);

/* H update: reads h[src], writes h[1-src] and h_t, partial sums to hpart.
 * write_h = 0 evaluates the loss pieces of the input state only. */
int espm_mu_step_h(const espm_mu_state* st, int src, int write_h, espm_stream_t stream);

/* Reduces hpart: history slot `slot` gets the loss pieces of h[src] and (slot > 0) rel_H of the update
 * that produced h[src]; hstat[1-src] gets the statistics of the new H (local; the caller max/sum-reduces
 * over ranks when sharded). */
int espm_mu_h_finalize(const espm_mu_state* st, int src, int slot, espm_stream_t stream);

/* Loss pieces of the state (w[src], h[src]) into history slot `slot`; H is not updated. */
int espm_mu_loss_only(const espm_mu_state* st, int src, int slot, espm_stream_t stream);

/* A_slab[b] = sum over the pixels of block b of R[:, j] H[:, j]^T with R = X / (GW H), H = h_t. */
int espm_mu_w_accum(const espm_mu_state* st, espm_stream_t stream);
/* espm_mu_step_h(st, src, 1) and espm_mu_w_accum as ONE launch where espm_mu_fused_applies(st) (see no_fused above), else
 * as those two calls: reads h[src]; writes h[1 - src], the records of hpart and a_slab.  (Fused: one record per block of
 * 1024 pixels in every other record slot, zeros - neutral for every field - in between, so that espm_mu_h_finalize and
 * the reduction calls read the records as they always do; h_t is left alone.) */
int espm_mu_step_hw(const espm_mu_state* st, int src, espm_stream_t stream);
int espm_mu_fused_applies(const espm_mu_state* st);
/* a = sum_b a_slab[b] in fixed order (one pass, bit-reproducible). */
int espm_mu_w_reduce(const espm_mu_state* st, espm_stream_t stream);
/* espm_mu_w_reduce and espm_mu_h_finalize(st, src, slot) in ONE launch (the reduction of the H-step's records
 * rides in an extra workgroup): for callers that sequence the half-steps themselves, after espm_mu_step_h(src, 1)
 * and espm_mu_w_accum. */
int espm_mu_w_reduce_finalize(const espm_mu_state* st, int src, int slot, espm_stream_t stream);
/* W update from a (sharded: the all-rank sum written by espm_mu_shard_combine) and hstat[hsrc] (global
 * row sums of the new H): reads w[src], writes w[1-src], gw_s, colsum_gw and rel_W into history slot `slot`. */
int espm_mu_w_finish(const espm_mu_state* st, int src, int hsrc, int slot, espm_stream_t stream);

/* The rest of the W-step after espm_mu_w_accum in one call: slab reduction (with espm_mu_h_finalize(st, src, slot)
 * riding in its launch when with_finalize != 0) and the W update w[src] -> w[1-src] with the row sums of the new H
 * h[1-src]; rel_W goes to history slot + 1.  When W' needs nothing global (G = identity, no simplex_W, n >= 64) the
 * reduction workgroups finish their own entries of W and a one-workgroup tail forms colsum(GW') and rel_W (w_scratch
 * holds the partials).  With the simplex over W, G = identity, every row in the simplex (no simplex_rows), the default
 * multiplicative rule and n_pad a multiple of 32, the multipliers follow from per-component sums over the channels
 * (dicotomy.py:111-173 applied to f(delta) = S / delta + n0 eps - 1): the reduction also leaves those sums per 32 channels
 * and a second many-workgroup launch finds the multipliers and updates W (same tail; no_fused = 1 keeps the one-workgroup
 * finish for A/B).  Otherwise this is espm_mu_w_reduce[_finalize] + espm_mu_w_finish. */
int espm_mu_w_reduce_finish(const espm_mu_state* st, int src, int slot, int with_finalize, espm_stream_t stream);

/* n_iter full iterations on one GPU, no host synchronisation; updates st->cur / st->it.
 * History slot t holds the loss pieces of state t (slot 0 = initial state) and the relative
 * changes of the update that produced it.  A final loss-only H-step fills the last slot when
 * final_loss != 0. */
int espm_mu_iterate(espm_mu_state* st, int n_iter, int final_loss, espm_stream_t stream);
/* 1 when the W update needs nothing global (G = identity, no simplex over W, ...): the slab / record reduction updates W
 * itself and a tail workgroup finishes (tail_mode above); 0: espm_mu_w_finish does the update, there is no tail. */
int espm_mu_w_update_is_local(const espm_mu_state* st);
/* The tail of the update w[src] -> w[1 - src] that produced state slot + 1, as a launch of its own (after a finish call
 * with ESPM_TAIL_DEFER when no H-step will carry it). */
int espm_mu_w_update_tail(const espm_mu_state* st, int src, int slot, espm_stream_t stream);

/* Sharded image (pixel rows split over ranks).  Per iteration every rank packs one
 * record [A | hstat of the new H | first and last owned image row of the new H], the caller
 * all-gathers the records (RCCL), and shard_combine sums A over ranks in fixed order, forms the
 * global row sums / maxima in hstat[hnew].  The halo rows for the next H-step are read in place
 * from the neighbours' records: offsets k*n_pad*4 + ESPM_HS_STRIDE*8 (+ k*ny*4 for the last row). */
size_t espm_mu_shard_record_bytes(const espm_mu_state* st);
int espm_mu_shard_pack(const espm_mu_state* st, int hnew, void* record, espm_stream_t stream);
/* espm_mu_w_reduce_finalize(st, src, slot) + espm_mu_shard_pack(st, 1 - src, record) in ONE launch: the slab reduction
 * writes A straight into the record, the finalize workgroup puts the statistics of the new H h[1-src] there (the
 * LOCAL ones; st->a and st->hstat[1-src] are not written) and copies its boundary rows. */
int espm_mu_w_reduce_pack(const espm_mu_state* st, int src, int slot, void* record, espm_stream_t stream);
int espm_mu_shard_combine(const espm_mu_state* st, const void* records, int world, int hnew,
                          espm_stream_t stream);
/* espm_mu_shard_combine(.., hnew = 1 - src) + espm_mu_w_finish(st, src, 1 - src, slot + 1) in one call; the sum over
 * the ranks and the W update share a launch when W' needs nothing global (see espm_mu_w_reduce_finish). */
int espm_mu_shard_combine_finish(const espm_mu_state* st, const void* records, int world, int src, int slot,
                                 espm_stream_t stream);

/* ---- one-shot record exchange between the ranks of a node (SURVEY 8b "espm_allreduce", 8e) --------------------------
 * Every rank owns a mailbox in uncached device memory, mapped into every peer through hipIpc (espm_xchg_handle ->
 * exchange the 64-byte handles by any means -> espm_xchg_connect).  espm_xchg_post copies the record staged at
 * espm_xchg_staging() into slot [seq & 1][rank] of EVERY rank's mailbox over the direct links and raises a flag there;
 * espm_xchg_wait returns (on the stream) once the flags of all ranks have reached seq - bounded: a peer that never
 * delivers is counted (espm_xchg_timeouts), not waited for.  espm_xchg_records(parity) is then what an all-gather would
 * have produced: world records in rank order.  Sequence numbers start at 1 and grow by one per exchange on every rank.
 * Contexts are opaque and owned by the library (the one allocation it makes: peers must be able to map it). */
typedef struct espm_xchg espm_xchg;
#define ESPM_XCHG_HANDLE_BYTES 64
int espm_xchg_create(int world, int rank, size_t record_bytes, espm_xchg** out);
int espm_xchg_handle(const espm_xchg* x, void* handle_out /* ESPM_XCHG_HANDLE_BYTES */);
int espm_xchg_connect(espm_xchg* x, const void* handles /* world * ESPM_XCHG_HANDLE_BYTES, rank order */);
void* espm_xchg_staging(const espm_xchg* x);
const void* espm_xchg_records(const espm_xchg* x, int parity);
int espm_xchg_post(espm_xchg* x, uint32_t seq, espm_stream_t stream);
int espm_xchg_wait(espm_xchg* x, uint32_t seq, espm_stream_t stream);
int espm_xchg_timeouts(const espm_xchg* x, uint32_t* count_out);   /* host-synchronous read of the give-up counter */
/* How a flag follows its record's stores: 0 (default) a relaxed system-scope store behind the drain of the write-through data
 * stores and the workgroup's barrier; 1 a release store at system scope.  The initial value is 1 when the environment holds
 * ESPM_XCHG_ORDER=release at espm_xchg_create.  Every rank must use the same order only in the sense that each protects its
 * OWN records; mixing is harmless.  (csrc/mu_xchg.hip: the ordering contract.) */
int espm_xchg_set_order(espm_xchg* x, int release);
int espm_xchg_order(const espm_xchg* x);
int espm_xchg_destroy(espm_xchg* x);

/* The sharded W step after the accumulation in ONE launch (where espm_mu_w_update_is_local; otherwise the four calls
 * espm_mu_w_reduce_pack -> espm_xchg_post -> espm_xchg_wait -> espm_mu_shard_combine_finish): every workgroup of the slab
 * reduction delivers its 32 entries of A to all ranks itself, waits for the same piece of every rank, sums them in rank
 * order and updates its entries of W; the statistics of the new H block go to every rank, its boundary rows to the
 * neighbours.  Pieces and statistics travel as 8-byte granules {value, sequence number} in an area of the mailbox of their
 * own (value and "it is there" in one store); afterwards the records of espm_xchg_records(x, seq & 1) hold what the NEXT
 * launch reads in place - the neighbours' boundary rows (and this rank's own statistics) - not the pieces of A, which went
 * straight into st->a.  The grid need not be resident at once: workgroups are dispatched in index order, each posts before it
 * waits, and waits only for the workgroup of its own index on the other ranks (csrc/mu_w_step.hip). */
int espm_mu_shard_exchange_finish(const espm_mu_state* st, espm_xchg* x, uint32_t seq, int src, int slot, espm_stream_t stream);

/* espm_mu_iterate with HIP events on the launch stream around the launches of every iteration (diagnostics: bench.py's roofline line).
 * first_ms[i]: iteration i's first launch (the H update; with the W accumulation where the fused kernel applies), rest_ms[i]: what
 * follows it up to the new W.  Host arrays of n_iter floats, 1 <= n_iter <= 4096; synchronises the stream before it returns.  The loop
 * is the one espm_mu_iterate runs (same launches, same order, enqueued from C: the device never waits for the host). */
int espm_mu_iterate_timed(espm_mu_state* st, int n_iter, float* first_ms, float* rest_ms, espm_stream_t stream);

/* n_iter iterations of a SHARDED image (pixel rows split over the ranks of x) without host synchronisation and without a
 * host-side collective: per iteration espm_mu_step_hw, espm_mu_w_reduce_pack into the staged record, espm_xchg_post /
 * _wait, espm_mu_shard_combine_finish on the gathered records; the halo rows of the next H-step are the neighbours'
 * records in the mailbox.  *seq is the exchange counter (in: last used, out: last used); every rank makes the same calls. */
int espm_mu_iterate_sharded(espm_mu_state* st, espm_xchg* x, uint32_t* seq, int n_iter, int final_loss, espm_stream_t stream);

/* nu (p) with sum_i max(num_ij / (nu_j + den_ij), log_shift) = 1.  num (k, p), den (k, den_cols)
 * with den_cols in {1, p}, fp64 device arrays.  status_out (device int32): number of columns
 * that violate the preconditions (dicotomy.py:17-19). */
int espm_dichotomy_simplex(const double* num, const double* den, int k, int p, int den_cols,
                           double log_shift, double tol, int maxit, double* nu_out,
                           int32_t* status_out, espm_stream_t stream);

/* The H update's per-pixel multiplier search as a launch of its own: the fp32 routine espm_mu_step_h / espm_mu_step_hw inline
 * (dicotomy.py:4-55 per column, solved in the shifted unknown delta = nu + min{den_i : num_i > 0}; csrc/mu_common.hpp: simplex_root).
 * num, den (k, p) fp32 device arrays -> delta_out (p), e_out (k, p) = the shifted denominators: the update is
 * max(num / (delta + e), log_shift).  fast_exit != 0: the confirming evaluation is left out where the Newton step's predicted
 * residual is inside tol (what the H update does).  status_out as in espm_dichotomy_simplex.  k within this library's range. */
int espm_simplex_root_f32(const float* num, const float* den, int k, int p, float log_shift, float tol, int maxit, int fast_exit,
                          float* delta_out, float* e_out, int32_t* status_out, espm_stream_t stream);

/* The other two multipliers of espm/estimators/dicotomy.py as module-level functions (fp64 device arrays, per-column
 * convergence): acc (dicotomy.py:57-82): sum_k max(sqrt((b_kj + nu_j)^2 + 4 a c_kj) - nu_j - b_kj, 2 a eps) = 2 a with b
 * (k, b_cols in {1, p}), minus_c (k, p) >= 0; pg (dicotomy.py:84-108): sum_k max(a_kj + nu_j, eps) = 1 with a (k, p). */
int espm_dichotomy_simplex_acc(double a, const double* b, const double* minus_c, int k, int p, int b_cols, double log_shift,
                               double tol, int maxit, double* nu_out, int32_t* status_out, espm_stream_t stream);
int espm_dichotomy_simplex_pg(const double* a, int k, int p, double log_shift, double tol, int maxit, double* nu_out,
                              espm_stream_t stream);

/* Frobenius ("l2") branch of the step functions, espm/estimators/updates.py:109-118 (H) and :31-36 (W); in the reference
 * reachable by calling multiplicative_step_h / _w with l2=True (espm/tests/test_updates.py:457-568) and, for the W step,
 * from a fit with algo="l2_surrogate", l2=True (smooth_nmf.py:404-413).  f32 store, xscale = 1; the H step also needs
 * lambda_L = 0, mu = NULL (the W step does not look at them).  work: (2, KP, KP) device floats (GW^T GW, then H H^T);
 * scratch: device doubles for the partial Gram sums (>= KP * KP, more = more workgroups).
 *   l2_step_h: H' = max(H * (GW^T X) / ((GW^T GW) H + nu), eps), simplex_h / fixed_h as in espm_mu_step_h; h[src] -> h[1-src].
 *   l2_step_w: W' = max(W / (G^T G W H H^T) * (G^T (X H^T)), eps), fixed_w; H = h_t (the caller keeps it current);
 *              gtg = G^T G (m, m) when G is given (a property of G alone: formed once by the caller); w[src] -> w[1-src]. */
int espm_mu_l2_step_h(const espm_mu_state* st, int src, float* work, double* scratch, int scratch_doubles, espm_stream_t stream);
int espm_mu_l2_step_w(const espm_mu_state* st, int src, const float* gtg, float* work, double* scratch, int scratch_doubles,
                      espm_stream_t stream);
/* The W step in two halves, for a sharded image: _partials leaves this rank's X H^T in st->a ((k, n_pad) fp32) and its
 * H H^T in work[KP*KP ..] (KP x KP fp32) - both plain sums over the rank's pixels, which the caller adds over the ranks -
 * and _finish forms W' from them.  espm_mu_l2_step_w is the two in a row. */
int espm_mu_l2_w_partials(const espm_mu_state* st, float* work, double* scratch, int scratch_doubles, espm_stream_t stream);
int espm_mu_l2_w_finish(const espm_mu_state* st, int src, const float* gtg, const float* work, espm_stream_t stream);

/* Terms of the linesearch on the Laplacian surrogate (espm/estimators/surrogates.py:65-149, smooth_nmf.py:376-381)
 * between two H buffers, Ht = h[hold] (before the update) and H = h[hnew]: out (4 + ESPM_KP device doubles) =
 * [sum Ht (Ht L), sum (Ht L) H, sum H (H L), sum (Ht - H)^2, dg_0 .. dg_7] with dg_k = sum_j Ht log(Ht / H) - Ht + H.
 * The caller forms d = 1/2 (2 out[1] - out[0] + gamma t3) - 1/2 out[2] with t3 = sum_k max_j H_kj dg_k (log surrogate,
 * Bregman) or t3 = out[3] (quadratic surrogate) - lambda_L = 1 as the reference calls it - and lowers gamma by 1.05
 * when d > 0, else raises it by 1.5.  Uses st->hpart as scratch: call it between
 * espm_mu_h_finalize and the next espm_mu_step_h.  One GPU only. */
int espm_mu_linesearch_terms(const espm_mu_state* st, int hold, int hnew, double* out, espm_stream_t stream);

/* The same for a rank of a sharded image: the terms of its block of image rows (the caller sums the `out` of the ranks, in
 * rank order).  The image rows above / below the block come from the neighbours' boundary rows: those of the new H from
 * st->halo_top / halo_bot, those of the H before the update from old_halo_top / old_halo_bot ((k, ny) fp32 each - the
 * previous exchange's records still hold them, the exchange being double buffered); null exactly where st's are. */
int espm_mu_linesearch_terms_sharded(const espm_mu_state* st, int hold, int hnew, const float* old_halo_top, const float* old_halo_bot,
                                     double* out, espm_stream_t stream);

/* The same terms without a state (module-level surrogates, espm/estimators/surrogates.py): h_old, h_new (k, ld) fp32 with p
 * used columns, grid (nx, ny) when grid_mode != 0 else L = identity; part: scratch of (4 + KP) * ceil(p / 512) doubles. */
int espm_surrogate_terms(const float* h_old, const float* h_new, int k, int p, int64_t ld, int nx, int ny, int grid_mode, double* part,
                         int part_doubles, double* out, espm_stream_t stream);

/* out = H @ L for the 5-point Laplacian on an (nx, ny) grid (k, nx*ny) with leading dim ld. */
int espm_mu_laplacian(const float* h, int k, int nx, int ny, int64_t ld, float* out,
                      espm_stream_t stream);

/* ---- initialisation on the device (host side: espm_amd/init_device.py) ----------------------------------
 * The LU normaliser of the randomized range finder behind the reference's NNDSVD initialisation
 * (espm/estimators/updates.py:179 -> sklearn _initialize_nmf -> _randomized_range_finder, power_iteration_normalizer "LU":
 * `Q, _ = scipy.linalg.lu(A, permute_l=True)`), called 2 * n_iter = 14 times per fit on matrices of n_components + 10 columns.
 * out (m, r) contiguous = P L of A = P L U, partial pivoting with LAPACK's pivot choice (largest modulus, first row on ties).
 * a: (m, r) with leading dimension ld, dtype ESPM_SRC_F32 / ESPM_SRC_F64 (out has the same); m >= r, r <= 64.
 * scratch: espm_lu_pl_scratch_bytes(m, r, dtype) bytes of device memory, contents irrelevant.  r + 1 launches on `stream`. */
size_t espm_lu_pl_scratch_bytes(int m, int r, int dtype);
int espm_lu_pl(const void* a, int dtype, int m, int r, int64_t ld, void* out, void* scratch, size_t scratch_bytes, espm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESPM_MU_H */
