"""ctypes binding of libespm_mu.so (include/espm_mu.h).

The HIP library is the product: there is NO CPU fallback.  A missing or stale library raises
at import of this module (``python -c "import __graft_entry__ as g; g.build()"`` rebuilds it).
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64); it must be loaded FIRST so that libespm_mu.so binds
# to the same runtime instance as the tensors and streams it is handed (two runtimes in one process do
# not share devices: "no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESPM_MU_LIB", os.path.join(_HERE, "lib", "libespm_mu.so"))

# mirrors of the header's constants
OK, EINVAL, ENOSOLUTION, EHIP, EUNSUPPORTED = 0, -1, -2, -3, -4
X_F32, X_BF16, X_U8, X_ELL = 0, 1, 2, 3
ELL_TILE, ELL_PB, ELL_PBITS, ELL_LDS_MAX = 512, 1024, 10, 144 * 1024
ELL_UNIT_ROWS, ELL_UNIT_MAX_N, ELL_PAIR_MAX_K = 8, 4096, 6
SRC_F32, SRC_F64 = 0, 1
LAYOUT_CM, LAYOUT_PM = 0, 1
MAX_K, KP, PPAD, NPAD = 8, 8, 512, 8          # (the default build; `variant(k)` below for the wide one)
HP_STRIDE, HS_STRIDE, HI_STRIDE = 24, 16, 8
HS_ROWSUM, HS_MAX = 0, 8
WIDE_MAX_K = 16
TAIL_DEFER, TAIL_RIDE = 1, 2
HI_KLX, HI_REG, HI_LAP, HI_SUMY, HI_BAD, HI_REL_W, HI_REL_H = 0, 1, 2, 3, 4, 5, 6

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class MUState(C.Structure):
    """struct espm_mu_state (same field order as include/espm_mu.h)."""
    _fields_ = [
        ("n", _i32), ("m", _i32), ("k", _i32), ("p", _i32), ("nx", _i32), ("ny", _i32),
        ("n_pad", _i32), ("p_pad", _i32), ("x_dtype", _i32), ("tile_px", _i32), ("nblk_w", _i32), ("x_tile", _i32), ("n_cm", _i32), ("h_variant", _i32),
        ("p_total", _i64),
        ("simplex_h", _i32), ("simplex_w", _i32), ("grid_mode", _i32), ("compute_loss", _i32),
        ("lambda_l", _f32), ("sigma_l", _f32), ("eps_reg", _f32), ("log_shift", _f32),
        ("dicotomy_tol", _f32), ("rel_tol", _f32), ("xscale", _f32), ("gw_floor", _f32),
        ("x_cm", _vp), ("x_pm", _vp), ("g", _vp), ("colsum_g", _vp),
        ("w", _vp * 2), ("gw_s", _vp), ("colsum_gw", _vp), ("gw_a", _vp), ("gw_p", _vp), ("h", _vp * 2), ("h_t", _vp),
        ("mu", _vp), ("fixed_h", _vp), ("fixed_w", _vp), ("simplex_rows", _vp),
        ("halo_top", _vp), ("halo_bot", _vp),
        ("hpart", _vp), ("hstat", _vp * 2), ("a_slab", _vp), ("a", _vp), ("w_scratch", _vp),
        ("hist", _vp), ("hist_len", _i32), ("cur", _i32), ("it", _i32),
        ("ell_h", _vp), ("ell_h_off", _vp), ("ell_klc", _vp), ("ell_w", _vp), ("ell_w_off", _vp), ("chan_perm", _vp),
        ("ell_cbits", _i32), ("n_cg", _i32), ("pix_perm", _vp), ("g_t", _vp), ("breg_sr_px", _vp), ("breg_sr_ch", _vp), ("h_rule", _i32), ("pg_gamma_w", _f32), ("pg_q", _vp),
        ("ell_fill_px", _vp), ("ell_fill_num", _vp), ("ell_fill_n", _i32), ("tail_mode", _i32),
    ]


# every symbol include/espm_mu.h declares: name -> (restype, argtypes)
_SP = C.POINTER(MUState)
SYMBOLS = {
    "espm_mu_version": (C.c_char_p, []),
    "espm_mu_last_error": (C.c_char_p, []),
    "espm_mu_query": (C.c_int, [_SP]),
    "espm_mu_w_update_is_local": (C.c_int, [_SP]),
    "espm_mu_w_update_tail": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_pack_x": (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_ell_count": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_plan": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_fill": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_hstat": (C.c_int, [_SP, C.c_int, _vp]),
    "espm_mu_build_gw": (C.c_int, [_SP, C.c_int, _vp]),
    "espm_mu_step_h": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_h_finalize": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_loss_only": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_w_accum": (C.c_int, [_SP, _vp]),
    "espm_mu_w_reduce": (C.c_int, [_SP, _vp]),
    "espm_mu_w_reduce_finalize": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_w_finish": (C.c_int, [_SP, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_iterate": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_shard_record_bytes": (C.c_size_t, [_SP]),
    "espm_mu_shard_pack": (C.c_int, [_SP, C.c_int, _vp, _vp]),
    "espm_mu_shard_combine": (C.c_int, [_SP, _vp, C.c_int, C.c_int, _vp]),
    "espm_mu_shard_combine_finish": (C.c_int, [_SP, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_w_reduce_finish": (C.c_int, [_SP, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_w_reduce_pack": (C.c_int, [_SP, C.c_int, C.c_int, _vp, _vp]),
    "espm_mu_linesearch_terms": (C.c_int, [_SP, C.c_int, C.c_int, _vp, _vp]),
    "espm_surrogate_terms": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "espm_dichotomy_simplex_acc": (C.c_int, [C.c_double, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "espm_dichotomy_simplex_pg": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp]),
    "espm_mu_l2_step_h": (C.c_int, [_SP, C.c_int, _vp, _vp, C.c_int, _vp]),
    "espm_mu_l2_step_w": (C.c_int, [_SP, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "espm_dichotomy_simplex": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "espm_mu_laplacian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i64, _vp, _vp]),
}


class EspmError(RuntimeError):
    """HIP runtime failure or unsupported configuration reported by libespm_mu."""


def _load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP library is required (no CPU fallback). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` at the repository root.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def last_error(which=None) -> str:
    return (which or lib).espm_mu_last_error().decode()


def check(rc: int, which=None) -> None:
    """Status code -> the exception the reference raises in the same situation."""
    if rc == OK:
        return
    msg = last_error(which)
    if rc == ENOSOLUTION:
        raise ValueError(msg)  # espm/estimators/dicotomy.py:22-23
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise EspmError(msg)


class Variant:
    """One build of the library (include/espm_mu.h): its handle and the sizes that follow its component stride KP."""

    def __init__(self, handle, kp, min_k, max_k):
        self.lib, self.KP, self.MIN_K, self.MAX_K = handle, kp, min_k, max_k
        self.HP_STRIDE, self.HS_STRIDE, self.HS_MAX = 8 + 2 * kp, 2 * kp, kp

    def check(self, rc):
        check(rc, self.lib)


_narrow = Variant(lib, KP, 1, MAX_K)
_wide = None
WIDE_LIB_PATH = os.path.join(_HERE, "lib", "libespm_mu_wide.so")


def variant(k) -> Variant:
    """The build that holds the kernels for k components: 1..8 libespm_mu.so, 9..16 libespm_mu_wide.so (dense stores)."""
    global _wide
    if k <= MAX_K:
        return _narrow
    if k > WIDE_MAX_K:
        raise NotImplementedError(f"n_components = {k}: the kernels are built for 1..{WIDE_MAX_K} components")
    if _wide is None:
        _wide = Variant(_load(WIDE_LIB_PATH), 16, MAX_K + 1, WIDE_MAX_K)
    return _wide
