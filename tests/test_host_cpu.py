"""CPU-only checks: the C ABI loads and exports what include/espm_mu.h declares, ctypes mirrors the
header, the host-side estimator logic (parameter coercions, API surface) matches the golden
captures, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from oracle import mu_oracle as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "espm_mu.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from espm_amd import _lib
    return _lib


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(espm_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared_functions()
    assert len(declared) >= 18
    for path in (lib.LIB_PATH, lib.WIDE_LIB_PATH, lib.WIDEST_LIB_PATH):   # the three builds of the sources: 1..8, 9..16 and 17..32 components
        cdll = C.CDLL(path)
        for name in declared:
            assert hasattr(cdll, name), f"{name} declared in espm_mu.h but not exported by {path}"
            assert name in lib.SYMBOLS, f"{name} has no ctypes prototype in espm_amd/_lib.py"
    assert sorted(lib.SYMBOLS) == declared
    assert b"gfx950" in lib.lib.espm_mu_version()
    wide = lib.variant(12)
    assert wide.lib is not lib.lib and (wide.KP, wide.HP_STRIDE, wide.HS_STRIDE, wide.HS_MAX) == (16, 40, 32, 16)
    assert lib.variant(8).lib is lib.lib
    widest = lib.variant(17)
    assert widest is lib.variant(32) and widest.lib is not wide.lib and (widest.KP, widest.HP_STRIDE, widest.HS_STRIDE, widest.HS_MAX) == (32, 72, 64, 32)
    assert widest.lib.espm_mu_state_size() == lib.lib.espm_mu_state_size()
    with pytest.raises(NotImplementedError):
        lib.variant(33)


def test_header_constants_and_struct_match_ctypes(lib):
    """The binding takes the layout of espm_mu_state and the sizes from include/espm_mu.h itself; here an independent
    (regex) reading of the header and the LIBRARY's own view (offsetof / sizeof of every field) must both agree with it."""
    text = open(HEADER).read()
    raw = dict(re.findall(r"#define\s+(ESPM_[A-Z_]+)\s+(\(?[-+*0-9A-Z_ ]+\)?)\s*(?:/\*|$)", text, flags=re.M))

    def value(name, depth=0):   # integer #defines, or arithmetic over other #defines (the sizes that follow ESPM_KP)
        expr = re.sub(r"ESPM_[A-Z_]+", lambda m: str(value(m.group(0), depth + 1)), raw[name])
        assert depth < 4 and re.fullmatch(r"[-+*0-9 ()]+", expr), (name, expr)
        return int(eval(expr))
    defs = {name: value(name) for name in raw}
    for name, val in (("MAX_K", lib.MAX_K), ("KP", lib.KP), ("PPAD", lib.PPAD), ("NPAD", lib.NPAD),
                      ("HP_STRIDE", lib.HP_STRIDE), ("HS_STRIDE", lib.HS_STRIDE), ("HI_STRIDE", lib.HI_STRIDE),
                      ("HS_MAX", lib.HS_MAX), ("HI_KLX", lib.HI_KLX), ("HI_REG", lib.HI_REG), ("HI_LAP", lib.HI_LAP),
                      ("HI_SUMY", lib.HI_SUMY), ("HI_BAD", lib.HI_BAD), ("HI_REL_W", lib.HI_REL_W),
                      ("HI_REL_H", lib.HI_REL_H), ("MU_ABI_VERSION", lib.ABI_VERSION)):
        assert int(defs["ESPM_" + name]) == val, name
    body = re.sub(r"/\*.*?\*/", "", text[text.index("typedef struct espm_mu_state {"):text.index("} espm_mu_state;")], flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*(?:\[\d+\])?\s*$", part.strip())[0])
    assert names == [f[0] for f in lib.MUState._fields_]
    assert names[:2] == ["struct_size", "abi_version"]
    from espm_amd import _abi
    for handle in (lib.lib, lib.variant(12).lib):   # both builds
        assert handle.espm_mu_state_size() == C.sizeof(lib.MUState) and handle.espm_mu_abi_version() == lib.ABI_VERSION
        assert handle.espm_mu_state_layout().decode() == _abi.layout_string(lib.MUState)


def test_a_drifted_layout_is_refused(lib):
    """A binding whose copy of the struct lost, gained or reordered a field must fail loudly, not have its pointers
    misread: every entry point checks struct_size / abi_version, and the loader compares the layouts field by field."""
    from espm_amd import _abi
    fields = list(lib.MUState._fields_)
    i = [n for n, _ in fields].index("x_cm")

    class Dropped(C.Structure):        # one pointer missing: everything behind it moves by 8 bytes
        _fields_ = fields[:i] + fields[i + 1:]

    class Swapped(C.Structure):        # same size, two pointers traded places
        _fields_ = fields[:i] + [fields[i + 1], fields[i]] + fields[i + 2:]

    st = Dropped()
    st.struct_size, st.abi_version = C.sizeof(Dropped), lib.ABI_VERSION
    st.n, st.p, st.k = 64, 512, 3
    raw = C.CDLL(lib.LIB_PATH)        # (untyped handle: the typed one would refuse the foreign struct before the call)
    raw.espm_mu_last_error.restype = C.c_char_p
    assert raw.espm_mu_query(C.byref(st)) == lib.EINVAL and b"drifted" in raw.espm_mu_last_error()
    assert raw.espm_mu_step_h(C.byref(st), 0, 1, None) == lib.EINVAL
    assert raw.espm_mu_iterate(C.byref(st), 1, 0, None) == lib.EINVAL
    good = lib.MUState()
    good.n, good.p, good.k = 64, 512, 3
    assert raw.espm_mu_query(C.byref(good)) == 0
    good.abi_version = lib.ABI_VERSION - 1
    assert raw.espm_mu_query(C.byref(good)) == lib.EINVAL
    # same size, wrong offsets: only the field-by-field comparison of the loader sees it
    assert C.sizeof(Swapped) == C.sizeof(lib.MUState)
    assert _abi.layout_string(Swapped) != lib.lib.espm_mu_state_layout().decode()


def test_integration_binding_is_generated_from_the_header():
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_integration.py"), "--check"]) == 0, \
        "INTEGRATION.md's espm_mu_state block is stale: run python tools/gen_integration.py"


def test_query_layout_without_gpu(lib):
    st = lib.MUState()
    st.n, st.p, st.k, st.x_dtype = 2048, 512 * 512, 5, lib.X_BF16
    assert lib.lib.espm_mu_query(C.byref(st)) == 0
    assert (st.n_pad, st.p_pad, st.tile_px) == (2048, 262144, 256)
    st.n, st.p = 1980, 128 * 128
    assert lib.lib.espm_mu_query(C.byref(st)) == 0
    assert st.n_pad == 1984 and st.p_pad == 16384 and st.tile_px == 128 and st.nblk_w >= 1
    st.n = 0
    assert lib.lib.espm_mu_query(C.byref(st)) == lib.EINVAL and b"must be >= 1" in lib.lib.espm_mu_last_error()


def test_argument_errors_map_to_reference_exceptions(lib):
    with pytest.raises(ValueError):     # dicotomy.py:22-23
        lib.check(lib.lib.espm_dichotomy_simplex(None, None, 3, 4, 4, 0.5, 1e-5, 100, None, None, None))
    st = lib.MUState()
    st.n, st.p, st.k, st.x_dtype = 8, 8, 9, 0
    lib.lib.espm_mu_query(C.byref(st))
    st.xscale = 1.0
    with pytest.raises(NotImplementedError):  # k > ESPM_MAX_K is refused, never silently degraded
        lib.check(lib.lib.espm_mu_step_h(C.byref(st), 0, 1, None))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from espm_amd.estimators import SmoothNMF
    from espm_amd.estimators.updates import multiplicative_step_h
    X = np.random.default_rng(0).random((6, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SmoothNMF(n_components=2, verbose=0).fit(X)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        multiplicative_step_h(X, np.eye(6), np.ones((6, 2)), np.ones((2, 8)))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "espm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_estimator_params_and_coercions(golden, capsys):
    from espm_amd.estimators import NMFEstimator, SmoothNMF
    g = golden("f8_api")
    est = SmoothNMF()
    ours = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)) for k, v in est.get_params().items()}
    assert ours == json.loads(str(g["default_params"]))
    e2 = SmoothNMF(simplex_H=True, simplex_W=True, l2=True, lambda_L=-1, algo="nope", epsilon_reg=0)
    assert dict(simplex_H=e2.simplex_H, simplex_W=e2.simplex_W, l2=e2.l2, lambda_L=e2.lambda_L, algo=e2.algo,
                epsilon_reg=e2.epsilon_reg) == json.loads(str(g["coerced"]))
    e3 = SmoothNMF(linesearch=True, lambda_L=0.0)
    assert dict(lambda_L=e3.lambda_L, linesearch=e3.linesearch) == json.loads(str(g["coerced_linesearch"]))
    assert "simplex constraint is applied to W and not to H" in capsys.readouterr().out
    assert issubclass(SmoothNMF, NMFEstimator)
    assert SmoothNMF.loss_names_ == ["KL_div_loss", "log_reg_loss", "Lapl_reg_loss", "gamma"]
    from sklearn.base import clone
    c = clone(SmoothNMF(n_components=4, lambda_L=2.0, mu=np.array([0.0, 1.0])))
    assert c.n_components == 4 and c.lambda_L == 2.0


def test_host_helpers_match_reference(golden):
    from espm_amd import utils
    from espm_amd.estimators.base import normalization_factor
    from espm_amd.estimators.updates import initialize_algorithms
    g4 = golden("f4_laplacian")
    for i, (nx, ny) in enumerate(g4["shapes"]):
        L = utils.create_laplacian_matrix(nx, ny)
        np.testing.assert_allclose(g4[f"s{i}_H"] @ L, g4[f"s{i}_HL"], rtol=1e-12, atol=1e-14)
        assert utils.classify_laplacian(L, nx * ny) == ("grid", (nx, ny))
    assert utils.classify_laplacian(utils.identity_laplacian(12), 12) == ("identity", None)
    with pytest.raises(NotImplementedError):
        utils.classify_laplacian(2 * utils.create_laplacian_matrix(3, 4), 12)
    g7 = golden("f7_init")
    _, W, H = initialize_algorithms(g7["X"], g7["G"], None, None, 3, "nndsvd", 0, True, False)
    np.testing.assert_allclose(W, g7["nndsvd_1_1_W"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(H, g7["nndsvd_1_1_H"], rtol=1e-9, atol=1e-14)
    D, Hr = utils.rescaled_DH(g7["G"] @ g7["Wgiven_W0"] * 3.0, g7["Hgiven_H0"] / 3.0)
    np.testing.assert_allclose(D, g7["rescaled_D"], rtol=1e-10)
    np.testing.assert_allclose(Hr, g7["rescaled_H"], rtol=1e-10)
    g8 = golden("f8_api")
    np.testing.assert_allclose(normalization_factor(g8["norm_X"], 5), g8["norm_factor"], rtol=1e-12)


def test_reference_loses_the_simplex_next_to_a_pole(golden):
    """Documents why the HIP path solves for nu + den_min: from an NNDSVD start (zeros clamped to 1e-14) the
    reference's first H update misses the simplex by 5e-3, the re-parametrised root by 1e-15."""
    X = golden("f8_api")["norm_X"]
    G, W0, H0 = oc.initialize_algorithms(X, None, None, None, 5, "nndsvd", 0, True, False)
    ref = oc.multiplicative_step_h(X, G, W0, H0, simplex_H=True)
    exact = oc.multiplicative_step_h(X, G, W0, H0, simplex_H=True, exact_root=True)
    assert np.abs(ref.sum(axis=0) - 1).max() > 1e-3
    assert np.abs(exact.sum(axis=0) - 1).max() < 1e-12
    ok = np.abs(ref.sum(axis=0) - 1) < 1e-4      # columns where the reference is well conditioned agree
    np.testing.assert_allclose(ref[:, ok], exact[:, ok], atol=2e-5)


def test_synthetic_generator_is_sharding_consistent():
    from espm_amd import synth
    full = synth.make_problem(32, 8, 6, 3, N=50.0, seed=1)
    part = synth.make_problem(32, 4, 6, 3, N=50.0, seed=1, row0=4, nx_total=8)
    np.testing.assert_allclose(part["weights"], full["weights"][4 * 6:], rtol=1e-12)
    np.testing.assert_allclose(full["weights"].sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(full["phases"].sum(axis=1), 1.0, rtol=1e-12)
    X = synth.sample_numpy(full, seed=0)
    assert X.shape == (32, 48) and (X >= 0).all() and X.sum() > 0


def test_lu_normaliser_of_the_device_init_equals_scipy():
    """espm_amd/init_device.py::_lu_pl (the power iterations' normaliser of scikit-learn's randomized SVD, which the reference
    initialises through: espm/estimators/updates.py:179) against scipy.linalg.lu(A, permute_l=True)[0] - tall, square and
    wide matrices, both precisions, and identical rows (pixels with the same few counts), which tie for a pivot."""
    import scipy.linalg as sl
    import torch
    from espm_amd.init_device import _lu_pl

    rs = np.random.RandomState(3)
    for shape in [(5000, 15), (257, 13), (15, 15), (12, 15), (40, 3)]:
        for dt, tol in ((np.float64, 5e-14), (np.float32, 5e-6)):
            A = rs.normal(size=shape).astype(dt)
            ref = sl.lu(A, permute_l=True)[0]
            got = _lu_pl(torch.from_numpy(A)).numpy()
            assert got.shape == ref.shape and got.dtype == ref.dtype
            np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
    A = rs.normal(size=(600, 15))
    A[100] = A[7] = A[431] = 50.0 * A[3]     # three identical rows, the largest of every column
    np.testing.assert_allclose(_lu_pl(torch.from_numpy(A)).numpy(), sl.lu(A, permute_l=True)[0], rtol=0, atol=5e-14)


@pytest.mark.parametrize("layout", ["cm", "pm"])
@pytest.mark.parametrize("fill_lines", [False, True])
@pytest.mark.parametrize("scale", [None, 0.37])
def test_host_copy_of_a_large_fit_equals_the_reference_passes(layout, fill_lines, scale):
    """estimators/base.py::_HostCopy (the estimator's own X_ of a large fit, made on worker threads) against the passes it
    replaces: remove_zeros_lines' copy with its filled lines (espm/estimators/base.py:519-528), then the normalisation
    (base.py:264-267) - values, dtype and memory order, in both ingest layouts."""
    from espm_amd.estimators.base import _HostCopy

    rs = np.random.RandomState(5)
    n, p = 37, 1000
    X = rs.poisson(0.4, size=(n, p)).astype(np.float32)
    X[5] = 0
    X[:, [3, 998]] = 0
    if layout == "pm":
        X = np.ascontiguousarray(X.T).T          # (n, p) view of a pixel-major array, what hspy_comp=True hands over
    zc, zp = X.sum(axis=1) == 0, X.sum(axis=0) == 0
    want = X.copy() if layout == "cm" else X.T.copy().T
    if fill_lines:
        want[:, zp] = 1e-14
        want[zc, :] = 1e-14
    if scale is not None:
        np.multiply(want, scale, out=want)
    hc = _HostCopy(X, layout)
    assert hc.shape == X.shape and hc.dtype == X.dtype
    hc.finish(zp if fill_lines else None, zc if fill_lines else None, 1e-14, scale)
    got = hc.result()
    assert got.dtype == want.dtype and got.flags.c_contiguous == want.flags.c_contiguous and got is not X
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(hc), want)
    cancelled = _HostCopy(X, layout)
    cancelled.cancel()
    assert cancelled.result() is None


def test_bench_line_keeps_the_drivers_contract():
    """bench.py (run by the driver on the GPU box): the flags it is called with and the keys of its one JSON line - checked in the
    source, since running it needs the device (profiles/r03ax_bench_default.json is a line it printed)."""
    import ast
    import json
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src, flag
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert f'"{key}":' in src, key
    assert 'out["cpu_baseline"]' in src or '"cpu_baseline"' in src
    line = json.loads(open(os.path.join(ROOT, "profiles", "r03ax_bench_default.json")).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and abs(line["ms_per_step"] * line["value"] / 1e3 - 1.0) < 1e-6


@pytest.mark.parametrize("init", ["nndsvd", "nndsvda", "nndsvdar"])
def test_nndsvd_with_the_long_factor_on_the_device_matches_sklearn_on_cpu_tensors(init):
    """espm_amd/init_device.py (round 5): the NNDSVD's post-processing of the long factor runs as tensor operations where the factor
    lives (`_nndsvd_long_on_device`); here on CPU tensors, against scikit-learn's `_initialize_nmf` (the reference's call,
    espm/estimators/updates.py:179) and against the host route it replaced (ESPM_INIT_NNDSVD=host)."""
    import torch
    from sklearn.decomposition._nmf import _initialize_nmf
    from espm_amd import init_device
    rng = np.random.default_rng(21)
    n, p, k = 40, 900, 4
    X = rng.poisson(3.0 * (rng.random((n, k)) ** 3) @ (rng.random((k, p)) ** 2)).astype(np.float64)
    Wr, Hr = _initialize_nmf(X, n_components=k, init=init, random_state=3)
    Wd, Hd = init_device.initialize_nmf_device(X, k, init=init, random_state=3, X_device=torch.from_numpy(X))
    assert Wd.dtype == Wr.dtype and Hd.dtype == Hr.dtype and Hd.shape == Hr.shape
    if init == "nndsvdar":   # (the random fill follows the zero pattern: compare where both routes decided the entry by the SVD)
        kept = (Hr > 1e-3 * Hr.max()) & (Hd > 1e-3 * Hd.max())
        np.testing.assert_allclose(Hd[kept], Hr[kept], rtol=1e-6, atol=1e-9)
    else:
        np.testing.assert_allclose(Wd, Wr, rtol=1e-6, atol=1e-9 * np.abs(Wr).max())
        np.testing.assert_allclose(Hd, Hr, rtol=1e-6, atol=1e-9 * np.abs(Hr).max())
    os.environ["ESPM_INIT_NNDSVD"] = "host"
    try:
        Wh, Hh = init_device.initialize_nmf_device(X, k, init=init, random_state=3, X_device=torch.from_numpy(X))
    finally:
        del os.environ["ESPM_INIT_NNDSVD"]
    np.testing.assert_allclose(Wd, Wh, rtol=1e-9, atol=1e-12)
    if init != "nndsvdar":
        np.testing.assert_allclose(Hd, Hh, rtol=1e-9, atol=1e-12)


def test_host_measures_and_generic_bisection_golden(golden):
    """Fixture F20, from the reference: the measures the path's own tests lean on (espm/measures.py: KL :427-454, KL_loss_surrogate :506-522 -
    here without the reference's n x k x p arrays -, log_surrogate :550-558, r2 / ordered_r2 :99-119, :315-327) and the generic bisection
    `dicotomy` (estimators/dicotomy.py:111-173: one stop rule for all entries, brackets updated in place).  Host functions: no GPU needed."""
    from espm_amd import measures as M
    from espm_amd.estimators.dicotomy import dicotomy
    g = golden("f20_measures_and_dicotomy")
    X, W, H, Ht, Y, mu = (g[k] for k in ("X", "W", "H", "Ht", "Y", "mu"))
    np.testing.assert_allclose(M.KL(X, Y), g["KL"], rtol=1e-12)
    np.testing.assert_allclose(M.KL(X, Y, average=True), g["KL_avg"], rtol=1e-12)
    np.testing.assert_allclose(M.KL_loss_surrogate(X, W, H, Ht), g["KLs"], rtol=1e-12)
    np.testing.assert_allclose(M.KL_loss_surrogate(X, W, H, Ht, average=True), g["KLs_avg"], rtol=1e-12)
    np.testing.assert_allclose(M.KL_loss_surrogate(X, W, Ht, Ht), g["KLs_at"], rtol=1e-12)
    # the majoriser touches the divergence where H = Ht (measures.py's test_surrogates; the factorised KL restated in numpy here)
    Xc, Yc = np.maximum(X, 1e-14), W @ Ht
    np.testing.assert_allclose(g["KLs_at"] - np.sum(Xc), np.sum(Yc) - np.sum(Xc) - np.sum(Xc * np.log(Yc)), rtol=1e-10)
    np.testing.assert_allclose(M.log_surrogate(H, Ht, mu, 0.8), g["logs"], rtol=1e-13)
    np.testing.assert_allclose(M.log_surrogate(H, Ht, mu, 0.8, average=True), g["logs_avg"], rtol=1e-13)
    np.testing.assert_allclose(M.log_surrogate(H, Ht, 0.3, 1.0), g["logs_scalar"], rtol=1e-13)
    np.testing.assert_allclose(M.r2(g["maps_t"][0], g["maps_a"][1]), g["r2"], rtol=1e-13)
    np.testing.assert_allclose(M.ordered_r2(g["maps_t"], g["maps_a"], [2, 0, 3, 1]), g["ordered_r2"], rtol=1e-13)
    t = g["dic_t"]
    lo, hi = np.zeros(9), np.full(9, 8.0)
    root = dicotomy(lo, hi, lambda x: np.exp(-x) * (t - x) + 0.1 * (t - x), 100, 1e-7)
    assert np.array_equal(root, g["dic_root"]) and np.array_equal(lo, g["dic_a"]) and np.array_equal(hi, g["dic_b"])
    with pytest.raises(AssertionError):
        dicotomy(np.zeros(3), np.ones(3), lambda x: x + 1.0, 10, 1e-6)     # no sign change: the reference's assertion

