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

from . import _abi

# ---- constants and the state layout come from include/espm_mu.h itself (espm_amd/_abi.py parses it): no hand-kept copy ----
_HEADER_TEXT = _abi.header_text()
_D = _abi.parse_defines(_HEADER_TEXT)
OK, EINVAL, ENOSOLUTION, EHIP, EUNSUPPORTED = 0, -1, -2, -3, -4          # enum espm_status
X_F32, X_BF16, X_U8, X_ELL = 0, 1, 2, 3                                  # enum ESPM_X_*
SRC_F32, SRC_F64 = 0, 1
LAYOUT_CM, LAYOUT_PM = 0, 1
ABI_VERSION = _D["ESPM_MU_ABI_VERSION"]
XCHG_HANDLE_BYTES = _D["ESPM_XCHG_HANDLE_BYTES"]
ELL_TILE, ELL_PB, ELL_PBITS, ELL_LDS_MAX = _D["ESPM_ELL_TILE"], _D["ESPM_ELL_PB"], _D["ESPM_ELL_PBITS"], _D["ESPM_ELL_LDS_MAX"]
ELL_STREAM_BYTES = _D["ESPM_ELL_STREAM_BYTES"]
ELL_BUCKETS = _D["ESPM_ELL_BUCKETS"]
ELL_UNIT_ROWS, ELL_UNIT_MAX_N, ELL_PAIR_MAX_K = _D["ESPM_ELL_UNIT_ROWS"], _D["ESPM_ELL_UNIT_MAX_N"], _D["ESPM_ELL_PAIR_MAX_K"]
KP, PPAD, NPAD = _D["ESPM_KP"], _D["ESPM_PPAD"], _D["ESPM_NPAD"]         # (the default build; `variant(k)` below for the wide one)
MAX_K = KP
HP_STRIDE, HS_STRIDE, HI_STRIDE = 8 + 2 * KP, 2 * KP, _D["ESPM_HI_STRIDE"]
HS_ROWSUM, HS_MAX = 0, KP
WIDE_MAX_K = 16      # libespm_mu_wide.so: 9..16 components (and the sparse count store's limit)
WIDEST_MAX_K = 32    # libespm_mu_wide32.so: 17..32 components, dense stores only
TAIL_DEFER, TAIL_RIDE = _D["ESPM_TAIL_DEFER"], _D["ESPM_TAIL_RIDE"]
HI_KLX, HI_REG, HI_LAP, HI_SUMY, HI_BAD, HI_REL_W, HI_REL_H = (_D["ESPM_HI_" + n] for n in ("KLX", "REG", "LAP", "SUMY", "BAD", "REL_W", "REL_H"))

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class MUState(C.Structure):
    """struct espm_mu_state: fields, order and types parsed from include/espm_mu.h.  A fresh instance carries
    struct_size / abi_version, which every entry point checks (ESPM_EINVAL on a mismatch)."""
    _fields_ = _abi.parse_struct(_HEADER_TEXT)

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))
        self.abi_version = ABI_VERSION


# every symbol include/espm_mu.h declares: name -> (restype, argtypes)
_SP = C.POINTER(MUState)
SYMBOLS = {
    "espm_mu_version": (C.c_char_p, []),
    "espm_mu_last_error": (C.c_char_p, []),
    "espm_mu_state_size": (C.c_size_t, []),
    "espm_mu_abi_version": (C.c_int, []),
    "espm_mu_state_layout": (C.c_char_p, []),
    "espm_mu_query": (C.c_int, [_SP]),
    "espm_mu_w_update_is_local": (C.c_int, [_SP]),
    "espm_mu_w_update_tail": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_pack_x": (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_ell_count": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_count_hist": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_fill_hist": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_plan": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_ell_fill": (C.c_int, [_SP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "espm_mu_hstat": (C.c_int, [_SP, C.c_int, _vp]),
    "espm_mu_build_gw": (C.c_int, [_SP, C.c_int, _vp]),
    "espm_mu_step_h": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_h_finalize": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_loss_only": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_w_accum": (C.c_int, [_SP, _vp]),
    "espm_mu_step_hw": (C.c_int, [_SP, C.c_int, _vp]),
    "espm_mu_fused_applies": (C.c_int, [_SP]),
    "espm_mu_w_reduce": (C.c_int, [_SP, _vp]),
    "espm_mu_w_reduce_finalize": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_w_finish": (C.c_int, [_SP, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_iterate_timed": (C.c_int, [_SP, C.c_int, _vp, _vp, _vp]),
    "espm_mu_iterate": (C.c_int, [_SP, C.c_int, C.c_int, _vp]),
    "espm_mu_shard_record_bytes": (C.c_size_t, [_SP]),
    "espm_mu_shard_pack": (C.c_int, [_SP, C.c_int, _vp, _vp]),
    "espm_mu_shard_combine": (C.c_int, [_SP, _vp, C.c_int, C.c_int, _vp]),
    "espm_mu_shard_combine_finish": (C.c_int, [_SP, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_w_reduce_finish": (C.c_int, [_SP, C.c_int, C.c_int, C.c_int, _vp]),
    "espm_mu_w_reduce_pack": (C.c_int, [_SP, C.c_int, C.c_int, _vp, _vp]),
    "espm_xchg_create": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "espm_xchg_handle": (C.c_int, [_vp, _vp]),
    "espm_xchg_connect": (C.c_int, [_vp, _vp]),
    "espm_xchg_staging": (_vp, [_vp]),
    "espm_xchg_records": (_vp, [_vp, C.c_int]),
    "espm_xchg_post": (C.c_int, [_vp, C.c_uint32, _vp]),
    "espm_xchg_wait": (C.c_int, [_vp, C.c_uint32, _vp]),
    "espm_xchg_timeouts": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "espm_xchg_set_order": (C.c_int, [_vp, C.c_int]),
    "espm_xchg_order": (C.c_int, [_vp]),
    "espm_xchg_destroy": (C.c_int, [_vp]),
    "espm_mu_shard_exchange_finish": (C.c_int, [_SP, _vp, C.c_uint32, C.c_int, C.c_int, _vp]),
    "espm_mu_iterate_sharded": (C.c_int, [_SP, _vp, C.POINTER(C.c_uint32), C.c_int, C.c_int, _vp]),
    "espm_mu_linesearch_terms": (C.c_int, [_SP, C.c_int, C.c_int, _vp, _vp]),
    "espm_mu_linesearch_terms_sharded": (C.c_int, [_SP, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "espm_surrogate_terms": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "espm_dichotomy_simplex_acc": (C.c_int, [C.c_double, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "espm_dichotomy_simplex_pg": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp]),
    "espm_mu_l2_step_h": (C.c_int, [_SP, C.c_int, _vp, _vp, C.c_int, _vp]),
    "espm_mu_l2_step_w": (C.c_int, [_SP, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "espm_mu_l2_w_partials": (C.c_int, [_SP, _vp, _vp, C.c_int, _vp]),
    "espm_mu_l2_w_finish": (C.c_int, [_SP, C.c_int, _vp, _vp, _vp]),
    "espm_simplex_root_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "espm_dichotomy_simplex": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "espm_mu_laplacian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i64, _vp, _vp]),
    "espm_lu_pl_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "espm_lu_pl": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i64, _vp, _vp, C.c_size_t, _vp]),
}


class EspmError(RuntimeError):
    """HIP runtime failure or unsupported configuration reported by libespm_mu."""


class LostPeerError(EspmError):
    """A sharded fit's one-shot record exchange gave up waiting for a peer: the iterates since then are not valid
    (MUEngine.history raises it on every rank at the same read-back; the estimator restarts on the collective transport)."""


def _load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP library is required (no CPU fallback). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` at the repository root.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype, fn.argtypes = res, args
    # the library's view of the state against the header's: size, version, and every field's name, offset and size
    theirs, ours = lib.espm_mu_state_layout().decode(), _abi.layout_string(MUState)
    if lib.espm_mu_state_size() != C.sizeof(MUState) or lib.espm_mu_abi_version() != ABI_VERSION or theirs != ours:
        raise ImportError(f"{path} was built from another include/espm_mu.h (ABI {lib.espm_mu_abi_version()} / {lib.espm_mu_state_size()} bytes "
                          f"against {ABI_VERSION} / {C.sizeof(MUState)}): rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
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
_widest = None
WIDE_LIB_PATH = os.environ.get("ESPM_MU_WIDE_LIB", os.path.join(_HERE, "lib", "libespm_mu_wide.so"))
WIDEST_LIB_PATH = os.environ.get("ESPM_MU_WIDEST_LIB", os.path.join(_HERE, "lib", "libespm_mu_wide32.so"))


def variant(k) -> Variant:
    """The build that holds the kernels for k components: 1..8 libespm_mu.so, 9..16 libespm_mu_wide.so, 17..32 libespm_mu_wide32.so
    (the dense stores, both contractions on the matrix cores)."""
    global _wide, _widest
    if k <= MAX_K:
        return _narrow
    if k > WIDEST_MAX_K:
        raise NotImplementedError(f"n_components = {k}: the kernels are built for 1..{WIDEST_MAX_K} components")
    if k <= WIDE_MAX_K:
        if _wide is None:
            _wide = Variant(_load(WIDE_LIB_PATH), 16, MAX_K + 1, WIDE_MAX_K)
        return _wide
    if _widest is None:
        _widest = Variant(_load(WIDEST_LIB_PATH), 32, WIDE_MAX_K + 1, WIDEST_MAX_K)
    return _widest
