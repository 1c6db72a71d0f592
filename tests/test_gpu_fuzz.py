"""Randomised parity sweep: every solver / option combination the GPU path accepts, on small random problems, against the
fp64 oracle (which is pinned to the reference by the golden fixtures).  Seeded - the same 36 cases every run
(ESPM_FUZZ_CASES=400 ESPM_FUZZ_WIDE_CASES=200 for a longer sweep: a handful of draws are skipped - the reference's own
bisection refuses them, or the drawn projected-gradient step sizes are unstable - and the rest pass)."""
import contextlib
import io

import numpy as np
import pytest

from oracle import mu_oracle as oc

pytestmark = pytest.mark.gpu


def _case(seed, wide=False):
    rng = np.random.default_rng((1000, 5000, 9000)[int(wide)] + seed)
    n = int(rng.integers(20, 260))
    nx, ny = int(rng.integers(3, 18)), int(rng.integers(3, 18))
    # 9..16: the second build of the library; 17..32 (wide = 2): the third
    k = int(rng.integers(17, 33)) if int(wide) == 2 else (int(rng.integers(9, 17)) if wide else int(rng.integers(1, 9)))
    p = nx * ny
    algo = ["log_surrogate", "log_surrogate", "bmd", "l2_surrogate", "projected_gradient"][seed % 5]
    use_G = algo != "bmd" and rng.random() < 0.45
    m = int(rng.integers(k, k + 14)) if use_G else None
    counts = rng.random() < 0.6            # integer counts (sparse / u8 store) or arbitrary non-negative data (fp32 store)
    H = rng.random((k, p)) ** 2 + 0.03
    H /= H.sum(axis=0, keepdims=True)
    if use_G:
        G = rng.random((n, m)) * (rng.random((n, m)) < 0.5) + 0.01
        W = rng.random((m, k)) * 30.0 / n
        D = G @ W
    else:
        G = None
        W = rng.random((n, k)) ** 3 * 100.0 / n + 1e-3
        D = W
    Y = D @ H
    X = rng.poisson(Y).astype(np.float64) if counts else Y * (0.5 + rng.random(Y.shape))
    X[X.sum(axis=1) == 0, 0] = 1.0
    X[0, X.sum(axis=0) == 0] = 1.0
    W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
    H0 = rng.random((k, p)) + 0.05
    H0 /= H0.sum(axis=0, keepdims=True)
    kw = dict(simplex_H=bool(rng.random() < 0.6), simplex_W=False, lambda_L=float(rng.choice([0.0, 0.3, 1.0, 2.5])),
              mu=0)
    if algo in ("log_surrogate", "bmd", "projected_gradient") and rng.random() < 0.4:
        kw["mu"] = rng.random(k) * 0.2
    if algo == "log_surrogate" and not kw["simplex_H"] and rng.random() < 0.5:
        kw["simplex_W"] = True
    if algo == "l2_surrogate" and not kw["simplex_H"] and rng.random() < 0.3:
        kw["simplex_W"] = True
    extra = {}
    if algo != "projected_gradient" and kw["lambda_L"] > 0 and rng.random() < 0.35 and not (k == 1 and kw["simplex_H"]):
        # (k = 1 on the simplex pins H to 1: the surrogate gap is exactly 0 and the sign the linesearch tests is rounding noise)
        extra["linesearch"] = True
    if algo == "projected_gradient":
        Gd = np.eye(n) if G is None else G
        L = oc.laplacian_matrix(nx, ny)
        gh = np.abs(oc.gradH(X, Gd, W0, H0, mu=kw["mu"], lambda_L=kw["lambda_L"], L=L)).max()
        gw = np.abs(oc.gradW(X, Gd, W0, H0)).max()
        extra["gamma"] = [float(gh / 0.05), float(gw / (0.2 * W0.mean()))]
    if rng.random() < 0.25 and kw["simplex_H"] is False and k >= 2:
        fH = -np.ones((k, p))
        fH[0, ::4] = 0.2
        extra["fixed_H"] = fH
    # empty lines (base.py:519-528: filled with log_shift by the reference; the sparse store keeps its lists empty): one draw
    # in four has channels, one in four pixels without a single count (a generator of its own: the draws above stay as they were)
    rz = np.random.default_rng(77000 + seed + 500 * int(wide))
    if rz.random() < 0.25:
        X[rz.choice(n, size=int(rz.integers(1, 6)), replace=False), :] = 0
    if rz.random() < 0.25 and algo != "bmd" and not (algo == "l2_surrogate" and kw["lambda_L"] == 0 and kw["simplex_H"]):
        # (that combination is the classic rule with the reference's own bisection, which the oracle restates without the exact
        #  root: next to the pole where an empty pixel's multiplier sits it is off by 1e-3, test_host_cpu.py::test_reference_loses_...)
        X[:, rz.choice(p, size=int(rz.integers(1, max(2, p // 10))), replace=False)] = 0
    return dict(X=X, G=G, W0=W0, H0=H0, k=k, shape=(nx, ny), algo=algo, kw=kw, extra=extra, counts=counts)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ESPM_FUZZ_CASES", "36"))))
def test_random_configuration_matches_oracle(seed):
    _run(_case(seed), seed)


@pytest.mark.parametrize("blocks", ["full", "small"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ESPM_FUZZ_CASES", "36"))))
def test_random_configuration_on_the_fused_kernel(seed, blocks, monkeypatch):
    """The same draws through the fused H update + W accumulation launch, which images of this size never get by themselves:
    "full" - the sparse store at its full geometry (512-pixel tiles, 1024-pixel blocks: sizes known at compile time), "small" -
    the images' own 128-pixel blocks (the variant with run-time sizes); every component count 1..8 (4, 3 and 2 segments per
    list group), dictionary G, mu, the Laplacian, fixed_H, simplex over H or W, empty channels and pixels."""
    c = _case(seed)
    if not c["counts"] or c["algo"] not in ("log_surrogate", "bmd"):
        pytest.skip("the fused kernel serves the sparse store with the default H rule")
    if blocks == "full":
        monkeypatch.setenv("ESPM_FORCE_ELL_TILE", "512")
    else:
        monkeypatch.setenv("ESPM_FUSED", "always")
    _run(c, seed, expect_fused=True)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ESPM_FUZZ_WIDE_CASES", "20"))))
def test_random_configuration_with_9_to_16_components_matches_oracle(seed):
    _run(_case(seed, wide=True), seed)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ESPM_FUZZ_WIDEST_CASES", "20"))))
def test_random_configuration_with_17_to_32_components_matches_oracle(seed):
    _run(_case(seed, wide=2), seed)


def _general_finish_cases(count):
    """(child process, ESPM_W_FINISH_GENERAL=1) the draws that reach a one-workgroup W finish, one line per draw."""
    import json, traceback
    for wide in (0, 1):
        for seed in range(count):
            c = _case(seed, wide=wide)
            if c["G"] is None and not c["kw"]["simplex_W"] and c["X"].shape[0] >= 64:
                continue   # G = identity without the simplex over W: the reduction workgroups finish W themselves
            try:
                _run(c, seed)
                status, msg = "passed", ""
            except pytest.skip.Exception as e:
                status, msg = "skipped", str(e)
            except Exception:
                status, msg = "failed", traceback.format_exc()[-1500:]
            print("CASE " + json.dumps(dict(seed=seed, wide=wide, algo=c["algo"], k=c["k"], status=status, msg=msg)), flush=True)


def test_random_configurations_on_the_general_w_finish():
    """The one-workgroup W finish that keeps nothing in registers (`w_finish_kernel`: a dictionary G or a simplex over W with more
    rows than the register-resident kernels hold; the only one of the 17..32-component build) - images of the fuzz's size never
    reach it by themselves.  It had no projected-gradient and no Bregman branch until round 5 (found by the 17..32 fuzz above).
    The knob is read once per process: ONE child process runs the draws of both narrower builds under ESPM_W_FINISH_GENERAL=1."""
    import json, os, subprocess, sys
    count = int(os.environ.get("ESPM_FUZZ_GENERAL_CASES", "30"))
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_fuzz as t; t._general_finish_cases(%d)" % (here, os.path.dirname(here), count)
    pr = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ESPM_W_FINISH_GENERAL="1"), capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0, pr.stderr[-3000:]
    cases = [json.loads(ln[5:]) for ln in pr.stdout.splitlines() if ln.startswith("CASE ")]
    failed = [c for c in cases if c["status"] == "failed"]
    passed = [c for c in cases if c["status"] == "passed"]
    assert not failed, "\n".join(f"seed {c['seed']} wide {c['wide']} {c['algo']} k={c['k']}: {c['msg']}" for c in failed[:5])
    assert len(passed) >= count // 3 and {"projected_gradient", "bmd"} <= {c["algo"] for c in passed} or count < 20, [(c["algo"], c["status"]) for c in cases]


def test_estimator_matches_the_reference_itself_on_the_fuzz_draws(golden):
    """Fixture F21 holds what the REFERENCE's SmoothNMF returns on 48 of the draws above (generated by tests/golden/make_golden.py f21): the
    HIP path against it directly, without the oracle in between.  (Under simplex_H the reference's multiplier is its bisection's, accurate to
    1e-5, where this path converges the root: the looser of the two tolerances.)"""
    import json
    from espm_amd.estimators import SmoothNMF
    g = golden("f21_reference_on_the_fuzz_draws")
    n_ok = 0
    for tag, status in json.loads(str(g["index"])):
        if status != "ok":
            continue
        wide, seed = int(tag[1]), int(tag.split("_s")[1])
        c = _case(seed, wide=wide)
        assert float(c["X"].sum()) == float(g[f"{tag}_x_sum"]), tag
        est = SmoothNMF(n_components=c["k"], G=c["G"], shape_2d=c["shape"], algo=c["algo"], tol=0, no_stop_criterion=True, max_iter=6,
                        verbose=0, **c["kw"], **c["extra"])
        with contextlib.redirect_stdout(io.StringIO()):
            est.fit_transform(c["X"], W=c["W0"].copy(), H=c["H0"].copy())
        loose = c["kw"]["simplex_H"]
        msg = f"{tag}: {c['algo']} k={c['k']} {c['kw']} {sorted(c['extra'])} store={est._engine.x_store}"
        np.testing.assert_allclose(est.losses_, g[f"{tag}_losses"], rtol=1e-4 if loose else 2e-5, err_msg=msg)
        np.testing.assert_allclose(est.H_, g[f"{tag}_H"], rtol=2e-3 if loose else 5e-4, atol=2e-4 if loose else 5e-5, err_msg=msg)
        Wr = g[f"{tag}_W"]
        np.testing.assert_allclose(est.W_, Wr, rtol=2e-3 if loose else 5e-4, atol=(2e-3 if loose else 5e-4) * np.abs(Wr).mean(), err_msg=msg)
        n_ok += 1
    assert n_ok >= 40


def _run(c, seed, expect_fused=False):
    from espm_amd.estimators import SmoothNMF
    iters = 6
    try:
        ref = oc.fit(c["X"], c["k"], G=c["G"], W=c["W0"].copy(), H=c["H0"].copy(), shape_2d=c["shape"], algo=c["algo"], tol=0,
                     no_stop_criterion=True, max_iter=iters, exact_root=(c["algo"] == "log_surrogate"), **c["kw"], **c["extra"])
    except AssertionError:
        # e.g. the projected gradient with k = 1 on the simplex: the reference's bracket has f = 0 at its end (dicotomy.py:141-144)
        pytest.skip("the reference's own bisection refuses this draw")
    if not np.isfinite(ref["losses"]).all():
        pytest.skip("the oracle itself diverges on this draw")
    if c["algo"] == "projected_gradient" and (np.diff(ref["losses"]) > 0).any():
        # step sizes 1 / gamma too long for this draw: the iterates jump (losses of 1e3, components thrown onto the clamp, after which
        # rescaled_DH's least squares is degenerate): rounding differences are amplified without bound
        pytest.skip("unstable projected-gradient step sizes in this draw")
    est = SmoothNMF(n_components=c["k"], G=c["G"], shape_2d=c["shape"], algo=c["algo"], tol=0, no_stop_criterion=True, max_iter=iters,
                    verbose=0, **c["kw"], **c["extra"])
    with contextlib.redirect_stdout(io.StringIO()):
        GW = est.fit_transform(c["X"], W=c["W0"].copy(), H=c["H0"].copy())
    if expect_fused and est._engine.x_store == "ell":
        import ctypes
        assert est._engine.lib.espm_mu_fused_applies(ctypes.byref(est._engine.st)) == 1
    tag = f"seed {seed}: {c['algo']} k={c['k']} G={'yes' if c['G'] is not None else 'no'} {c['kw']} {sorted(c['extra'])} store={est._engine.x_store}"
    # the reference's bisections stop at 1e-5 (global rule); the oracle uses the exact root only for the default solver
    loose = c["algo"] != "log_surrogate" and c["kw"]["simplex_H"]
    np.testing.assert_allclose(est.losses_, ref["losses"], rtol=1e-4 if loose else 2e-5, err_msg=tag)
    det = np.array(est.detailed_losses_, dtype=float)
    np.testing.assert_allclose(det[:, 3], ref["detailed_losses"][:, 3], rtol=1e-9, err_msg=tag + " (gamma)")
    np.testing.assert_allclose(est.H_, ref["H"], rtol=2e-3 if loose else 5e-4, atol=2e-4 if loose else 5e-5, err_msg=tag)
    np.testing.assert_allclose(GW, ref["GW"], rtol=2e-3 if loose else 5e-4, atol=(2e-3 if loose else 5e-4) * np.abs(ref["GW"]).mean(),
                               err_msg=tag)
