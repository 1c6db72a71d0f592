"""The hyperspy side of the drop-in (espm_amd/hyperspy_adapter.py) - hyperspy itself is not installable here, so the
contract is driven through the adapter's own signal class: what hyperspy's decomposition does with a custom algorithm
object (fit_transform on (pixels, channels), components_ -> factors, the estimator kept in learning_results) and what the
reference's EDSespm adds (X, shape_2d; espm/datasets/eds_spim.py:113-139, :484-604; espm/tests/test_datasets.py:235-260)."""
import contextlib
import io
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


class RecordingEstimator:
    """What hyperspy requires of a custom algorithm, recording what it was handed."""
    hspy_comp = True
    shape_2d = None

    def fit_transform(self, X):
        self.seen = X
        p, n = X.shape
        self.components_ = np.arange(2 * n, dtype=float).reshape(2, n)
        return np.ones((p, 2))


def test_signal_contract_without_a_gpu():
    from espm_amd import hyperspy_adapter as ha
    cube = np.arange(4 * 5 * 6, dtype=np.float64).reshape(4, 5, 6)
    s = ha.SpectrumImage(cube)
    assert s.shape_2d == (4, 5) and s.X.shape == (6, 20)
    assert np.shares_memory(s.X, cube) and np.shares_memory(s.unfolded(), cube)      # views: nothing is copied on the way in
    np.testing.assert_array_equal(s.X[:, 7], cube[1, 2])                              # pixel index = row * nx + column (utils.py:60-75)
    est = RecordingEstimator()
    lr = ha.decompose(s, est)
    assert est.shape_2d == (4, 5)                                                      # taken from the signal
    assert est.seen.shape == (20, 6) and np.shares_memory(est.seen, cube) and est.seen.flags.c_contiguous
    assert lr.decomposition_algorithm is est and lr.loadings.shape == (20, 2) and lr.factors.shape == (6, 2)
    assert s.get_decomposition_loadings().shape == (2, 4, 5) and s.get_decomposition_factors().shape == (2, 6)
    est.hspy_comp = False
    with pytest.raises(ValueError, match="hspy_comp"):
        ha.decompose(s, est)
    with pytest.raises(ValueError):
        ha.SpectrumImage(np.zeros((3, 4)))
    assert ha.register() in (True, False)          # True only where the reference package is importable


def test_extension_declaration_names_an_importable_class():
    import yaml
    with open(os.path.join(ROOT, "espm_amd", "hyperspy_extension.yaml")) as f:
        ext = yaml.safe_load(f)
    (name, spec), = ext["signals"].items()
    import importlib
    mod = importlib.import_module(spec["module"])
    assert hasattr(mod, name) and spec["signal_type"] == "EDS_espm_amd"
    text = open(os.path.join(ROOT, "pyproject.toml")).read()
    assert '[project.entry-points."hyperspy.extensions"]' in text and 'espm_amd = "espm_amd"' in text


@pytest.mark.gpu
def test_decomposition_end_to_end_matches_a_direct_fit():
    """decomposition(algorithm=est) on a cube == est.fit_transform on the (n, p) matrix: loadings = H^T, factors = G W, the
    estimator (with W_, G_, H_) in learning_results - and a large cube goes to the device pixel-major, as it lies."""
    from espm_amd import hyperspy_adapter as ha, synth
    from espm_amd.estimators import NMFEstimator, SmoothNMF
    n, ny, nx, k = 96, 14, 12, 3
    prob = synth.make_problem(n, ny, nx, k, N=120.0, seed=5)
    X = synth.sample_numpy(prob, seed=5)                                   # (n, p)
    cube = np.ascontiguousarray(X.T).reshape(ny, nx, n)
    W0, H0 = synth.random_init(n, k, ny * nx, seed=5, scale=0.2)
    kw = dict(n_components=k, lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0, no_stop_criterion=True, max_iter=25, verbose=0)
    direct = SmoothNMF(shape_2d=(ny, nx), **kw)
    GW = quiet(direct.fit_transform, X, W=W0.copy(), H=H0.copy())

    class Seeded(SmoothNMF):                                               # hyperspy passes only the data: fix the initial state
        def fit_transform(self, Xp, y=None, W=None, H=None):
            return super().fit_transform(Xp, W=W0.copy(), H=H0.copy())
    est = Seeded(hspy_comp=True, **kw)
    s = ha.SpectrumImage(cube)
    lr = quiet(ha.decompose, s, est)
    assert isinstance(lr.decomposition_algorithm, NMFEstimator) and est.shape_2d == (ny, nx)
    np.testing.assert_allclose(lr.loadings, direct.H_.T, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lr.factors, GW, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(est.losses_, direct.losses_, rtol=1e-9)
    np.testing.assert_allclose((est.G_ @ est.W_ @ est.H_).sum(axis=1), s.X.sum(axis=1), rtol=0.5, atol=1.0)   # test_datasets.py:256 (+ channels without counts)
    # a cube large enough for the device-side preparation: (pixels, channels) is uploaded as it is
    n2, ny2, nx2 = 512, 96, 96
    prob2 = synth.make_problem(n2, ny2, nx2, k, N=200.0, seed=6)
    import torch
    Xp = synth.sample_torch(prob2, "cuda", seed=6).cpu().numpy()           # (p, n) float32
    big = ha.SpectrumImage(Xp.reshape(ny2, nx2, n2))
    est2 = SmoothNMF(hspy_comp=True, n_components=k, lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0, no_stop_criterion=True,
                     max_iter=10, verbose=0, init="nndsvda", random_state=0)
    quiet(ha.decompose, big, est2)
    assert est2._ingest_layout == "pm" and est2.shape_2d == (ny2, nx2)
    ref = SmoothNMF(shape_2d=(ny2, nx2), n_components=k, lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0, no_stop_criterion=True,
                    max_iter=10, verbose=0, init="nndsvda", random_state=0)
    quiet(ref.fit_transform, np.ascontiguousarray(Xp.T))
    assert ref._ingest_layout == "cm"
    np.testing.assert_allclose(est2.losses_, ref.losses_, rtol=2e-6)
    np.testing.assert_allclose(est2.H_, ref.H_, rtol=1e-2, atol=3e-3)   # (the device NNDSVD multiplies X in the other stride order: rounding-level differences in W0, H0)
    assert np.all(np.diff(est2.losses_) < 0)
    del torch


@pytest.mark.gpu
def test_pixel_major_ingest_against_the_reference_fixture():
    """Fixture F17 (tests/golden/make_golden.py::f17): the REFERENCE's SmoothNMF(hspy_comp=True) on the (pixels, channels)
    matrix of a 96 x 96 x 512 cube (espm/estimators/base.py:243-247, :412-420; eds_spim.py:597-604).  Here the same cube goes
    through hyperspy's calling convention - SpectrumImage.decomposition -> fit_transform((p, n)) - at a size where the matrix
    is uploaded as it lies (pixel-major ingest) and everything the reference returns or leaves on the estimator is compared
    with what the reference produced: loadings, components_, W_, H_, losses_, rel_, n_iter_."""
    from espm_amd import hyperspy_adapter as ha
    from espm_amd.estimators import SmoothNMF
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "f17_hyperspy_ingest.npz")))
    nx, ny = (int(v) for v in g["shape"])
    Xp = g["X_u8"].astype(np.float64)                                        # (p, n), as hyperspy unfolds the cube
    p, n = Xp.shape
    k = g["W0"].shape[1]
    cube = Xp.reshape(nx, ny, n)

    class Seeded(SmoothNMF):                                                 # hyperspy passes only the data: fix the initial state
        def fit_transform(self, Xq, y=None, W=None, H=None):
            assert Xq.shape == (p, n) and np.shares_memory(Xq, cube)         # the cube's own memory, not a transposed copy
            return super().fit_transform(Xq, W=g["W0"].copy(), H=g["H0"].copy())

    for tag, kw, iters in (("free", dict(tol=0, no_stop_criterion=True), 30), ("stop", dict(tol=6e-4), 200)):
        est = Seeded(n_components=k, hspy_comp=True, simplex_H=True, simplex_W=False, lambda_L=1.0, verbose=0, max_iter=iters, **kw)
        s = ha.SpectrumImage(cube)
        lr = quiet(ha.decompose, s, est)                                     # shape_2d comes from the signal
        assert est._ingest_layout == "pm" and est.shape_2d == (nx, ny)
        assert est._engine.x_store == "ell"                                  # count data: the sparse store
        assert est.n_iter_ == int(g[f"{tag}_n_iter"]), (tag, est.n_iter_)
        np.testing.assert_allclose(est.losses_, g[f"{tag}_losses"], rtol=1e-5)          # (measured ~1e-7)
        np.testing.assert_allclose(lr.loadings, g[f"{tag}_loadings"], rtol=0, atol=5e-5)
        scale = np.abs(g[f"{tag}_components"]).max()
        np.testing.assert_allclose(lr.factors.T, g[f"{tag}_components"], rtol=0, atol=2e-4 * scale)
        np.testing.assert_allclose(est.components_, g[f"{tag}_components"], rtol=0, atol=2e-4 * scale)
        np.testing.assert_allclose(est.H_, g[f"{tag}_H"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(est.W_, g[f"{tag}_W"], rtol=0, atol=2e-4 * np.abs(g[f"{tag}_W"]).max())
        np.testing.assert_allclose(np.array(est.rel_), g[f"{tag}_rel"], rtol=2e-3, atol=1e-6)
        assert lr.loadings.shape == (p, k) and lr.factors.shape == (n, k) and lr.decomposition_algorithm is est
        np.testing.assert_array_equal(est.X_.shape, (n, p))
