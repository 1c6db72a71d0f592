"""GPU parity of the update rules: HIP path (through the C ABI) vs the numpy oracle and the golden
vectors captured from the reference; plus the reference's own property tests
(espm/tests/test_updates.py:93-249, :439-574, espm/tests/test_laplacian.py, test_measures.py:154-221)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mu_oracle as oc  # noqa: E402

F32 = dict(rtol=2e-5, atol=2e-6)


@pytest.fixture(scope="module")
def api():
    from espm_amd import measures, utils
    from espm_amd.estimators import dicotomy, updates
    return dict(step_h=updates.multiplicative_step_h, step_w=updates.multiplicative_step_w,
                dicho=dicotomy.dichotomy_simplex, init=updates.initialize_algorithms, measures=measures, utils=utils)


# ------------------------------------------------------------------------------------ dichotomy
def test_dichotomy_golden(api, golden):
    g = golden("f1_dichotomy")
    for i in range(int(g["n_kat"])):
        num, den, eps, tol = g[f"kat{i}_num"], g[f"kat{i}_den"], float(g[f"kat{i}_eps"]), float(g[f"kat{i}_tol"])
        nu = api["dicho"](num, den, eps, tol=tol)
        np.testing.assert_allclose(nu, g[f"kat{i}_nu"], atol=4 * max(tol, 1e-12) * max(1, np.abs(g[f"kat{i}_nu"]).max()))
    for c in range(int(g["n_rnd"])):
        num, den, eps = g[f"rnd{c}_num"], g[f"rnd{c}_den"], float(g[f"rnd{c}_eps"])
        nu = api["dicho"](num, den, eps, tol=0.0, maxit=100)
        ref = g[f"rnd{c}_t0_nu"]  # reference iterated to machine precision (100 sweeps)
        np.testing.assert_allclose(nu, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max(), err_msg=f"case {c}")
        f = np.sum(np.maximum(num / (den + nu), eps), axis=0) - 1
        assert np.abs(f).max() < 1e-7


def test_dichotomy_reference_properties(api):
    """espm/tests/test_updates.py:93-183."""
    rng = np.random.default_rng(1)
    tol = 1e-8
    num, den = rng.random((1, 1)) + 1, rng.random((1, 1))
    assert abs(api["dicho"](num, den, 0, tol=tol) - (num - den)) < 2 * tol
    n = 10
    num, den = rng.random((1, n)), rng.random((1, n))
    np.testing.assert_allclose(api["dicho"](num, den, 0, tol=tol), np.squeeze(num - den), atol=tol)
    num, den = rng.random((n, 6)), rng.random((n, 6))
    sol = api["dicho"](num, den, 0, tol=1e-6)
    np.testing.assert_allclose(np.sum(num / (den + sol), axis=0), np.ones(6), atol=1e-6)
    for den in (np.array([[1, 1, 3, 5, 4, 2]], float).T, np.array([[1, 1, 0, 5, 0, 2]], float).T):
        num = np.array([[1, 1, 0, 0, 0, 2]], float).T
        sol = api["dicho"](num, den, 0, tol=1e-6)
        np.testing.assert_allclose(np.sum(num / (den + sol)), 1, atol=1e-6)
    sol = api["dicho"](np.array([[3, 0.5]]).T, np.array([[1.0, 1]]).T, 1 / 4, tol=tol)
    assert abs(sol - 3) < 2 * tol
    with pytest.raises(ValueError):
        api["dicho"](rng.random((1, n)), rng.random((1, n)), 1.1, tol=tol)
    with pytest.raises(ValueError):
        api["dicho"](rng.random((3, n)), rng.random((3, n)), 0.5, tol=tol)
    num, den = rng.random((n, 6)), rng.random((n, 6))
    sol = api["dicho"](num, den, 0.05, tol=1e-6)
    np.testing.assert_allclose(np.sum(np.maximum(num / (den + sol), 0.05), axis=0), np.ones(6), atol=1e-6)
    with pytest.raises(AssertionError):
        api["dicho"](np.zeros((3, 4)), np.ones((3, 4)), 0.0)


def test_dichotomy_twelve_decades(api):
    """espm/tests/test_updates.py:185-249 (k=5, p=6400, scales 1e-6..1e6, with zeros and (k,1) denominators)."""
    k, p = 5, 6400
    span = np.logspace(-6, 6, num=17)
    rng = np.random.default_rng(0)
    for zeros in (False, True):
        for den_cols in (p, 1):
            num = rng.choice(span, (k, p)) * rng.random((k, p))
            if zeros:
                num[np.tile(np.arange(k), p // k), np.arange(p)] = 0
            den = rng.choice(span, (k, den_cols)) * rng.random((k, den_cols))
            for eps in (0.0, 0.1 / k):
                sol = api["dicho"](num, den, eps, tol=0, maxit=100)
                v = np.sum(np.maximum(num / (den + sol), eps), axis=0)
                # a root next to a pole (nu ~ -den_i) is only representable to ulp(nu)/(nu + den_i) in f:
                # the reference's own test accepts 1e-2 here
                np.testing.assert_allclose(v, np.ones(p), atol=1e-4)
                assert np.mean(np.abs(v - 1) < 1e-9) > 0.99
                np.testing.assert_allclose(sol, oc.dichotomy_simplex(num, den, eps, tol=0, maxit=100), rtol=1e-5,
                                           atol=1e-12)


def _per_pixel_root(num, den, eps, tol, fast_exit):
    """The H update's own fp32 routine (simplex_root<float, K>) through the C ABI; returns f = sum max(num / (delta + e), eps) - 1
    evaluated in fp64 from its outputs, and delta."""
    import torch
    from espm_amd import _lib
    k, p = num.shape
    dev = torch.device("cuda", 0)
    d_num = torch.as_tensor(np.ascontiguousarray(num, np.float32), device=dev)
    d_den = torch.as_tensor(np.ascontiguousarray(den, np.float32), device=dev)
    d_delta = torch.empty(p, dtype=torch.float32, device=dev)
    d_e = torch.empty((k, p), dtype=torch.float32, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.espm_simplex_root_f32(d_num.data_ptr(), d_den.data_ptr(), k, p, float(eps), float(tol), 100, int(fast_exit),
                                              d_delta.data_ptr(), d_e.data_ptr(), d_status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(d_status.item()) == 0
    delta, e = d_delta.cpu().numpy().astype(np.float64), d_e.cpu().numpy().astype(np.float64)
    n32 = np.asarray(num, np.float32).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(n32 > 0, n32 / (delta[None, :] + e), 0.0)
    return np.sum(np.maximum(t, eps), axis=0) - 1.0, delta


def _near_pole_inputs(k, p, seed):
    """Columns whose root sits next to a pole: one tiny numerator on the smallest denominator, the rest of the mass just short of 1
    (the case ADVICE r4 found residuals of 7.6e-6 in with a fixed |f| <= 1e-4 exit)."""
    rng = np.random.default_rng(seed)
    den = rng.uniform(0.5, 30.0, (k, p))
    num = rng.uniform(0.0, 1.0, (k, p)) * den / k
    j = np.arange(p)
    small = rng.integers(0, k, p)
    den[small, j] = den.min(axis=0) * rng.uniform(0.2, 0.9, p)
    num[small, j] = 10.0 ** rng.uniform(-14, -3, p)
    # scale the other numerators so that sum num / (den - d*) is a little below / above 1: the root then lies within ~num_small of the pole
    dstar = den[small, j]
    others = num.copy()
    others[small, j] = 0.0
    s = np.sum(others / (den - dstar[None, :] + (den == dstar[None, :])), axis=0)
    others *= (rng.uniform(0.9, 1.0 - 1e-5, p) / np.maximum(s, 1e-30))[None, :]
    others[small, j] = num[small, j]
    return others, den


@pytest.mark.parametrize("k", [3, 5, 8])
def test_per_pixel_root_meets_its_tolerance_with_and_without_the_fast_exit(k):
    """the exit without the confirming evaluation must leave |f| inside the tolerance that evaluation enforces
    (min(tol, 1e-6) in the H update; the reference's dicotomy_tol is 1e-5, dicotomy.py:152)."""
    tol = 1e-6
    rng = np.random.default_rng(k)
    cases = [_near_pole_inputs(k, 20000, 10 + k)]
    den = rng.uniform(0.1, 5.0, (k, 20000))
    cases.append((rng.uniform(0.0, 1.0, (k, 20000)) * den * rng.uniform(0.2, 3.0, (1, 20000)), den))   # ordinary columns
    span = np.logspace(-6, 6, 17)
    cases.append((rng.choice(span, (k, 20000)) * rng.random((k, 20000)), rng.choice(span, (k, 20000)) * rng.random((k, 20000))))
    for num, den in cases:
        f0, d0 = _per_pixel_root(num, den, 1e-14, tol, fast_exit=0)
        f1, d1 = _per_pixel_root(num, den, 1e-14, tol, fast_exit=1)
        # fp32 resolution of f next to a pole: ulp(delta + e) / (delta + e) per term, i.e. a few 1e-7 on a sum of 1
        assert np.abs(f0).max() <= tol + 4e-7, np.abs(f0).max()
        assert np.abs(f1).max() <= tol + 4e-7, np.abs(f1).max()
        assert np.mean(d0 != d1) < 0.5   # (the exit changes which iterate is returned for some columns, never the tolerance)
    # a large log_shift (the reference's tests use 0.02): the clamp's kinks are not in the prediction, the exit must not be taken
    num, den = cases[1]
    f0, d0 = _per_pixel_root(num, den, 0.02, tol, fast_exit=0)
    f1, d1 = _per_pixel_root(num, den, 0.02, tol, fast_exit=1)
    np.testing.assert_array_equal(d0, d1)
    assert np.abs(f1).max() <= tol + 4e-7


# ------------------------------------------------------------------------------------ H step
def test_step_h_golden_grid(api, golden):
    g = golden("f2_step_h")
    nx, ny = g["shape_2d"]
    L = api["utils"].create_laplacian_matrix(nx, ny)
    for c in range(int(g["n_cases"])):
        t, simplex, lam, mu_on, fix_on = g[f"c{c}_cfg"]
        t = int(t)
        Hn = api["step_h"](g[f"in{t}_X"], g[f"in{t}_G"], g[f"in{t}_W"], g[f"in{t}_H"].copy(), simplex_H=bool(simplex),
                           mu=g["mu_vec"] if mu_on else 0, epsilon_reg=float(g["epsilon_reg"]), lambda_L=float(lam),
                           L=L, fixed_H=g[f"in{t}_fixed"] if fix_on else None)
        # the reference's own multiplier is only converged to dicotomy_tol = 1e-5 (global stop rule)
        tol = dict(rtol=3e-5, atol=3e-6) if simplex else F32
        np.testing.assert_allclose(Hn, g[f"c{c}_H"], err_msg=f"case {c}", **tol)


def test_step_h_golden_special(api, golden):
    g = golden("f2_step_h")
    nx, ny = g["shape_2d"]
    L = api["utils"].create_laplacian_matrix(nx, ny)
    for t in (0, 1):
        X, G, W, H = (g[f"in{t}_{s}"] for s in ("X", "G", "W", "H"))
        np.testing.assert_allclose(api["step_h"](X, G, W, H.copy(), simplex_H=True, mu=0.4, lambda_L=0.5, L=L, sigmaL=11.0),
                                   g[f"smu_{t}_H"], rtol=3e-5, atol=3e-6)
        np.testing.assert_allclose(api["step_h"](X, G, W, H.copy(), simplex_H=True, lambda_L=1.5,
                                                 L=api["utils"].identity_laplacian(nx * ny)), g[f"lid_{t}_H"],
                                   rtol=3e-5, atol=3e-6)
    with pytest.raises(ValueError):
        api["step_h"](X, G, W, H, lambda_L=1.0, L=None)
    with pytest.raises(NotImplementedError):
        api["step_h"](X, G, W, H, use_bregman=True)   # G is a dictionary here: the Bregman variant needs G = identity


def test_frobenius_branch_golden_and_properties(api, golden):
    """l2=True: the Frobenius branch of both step functions (updates.py:31-36, :109-118) against the reference's outputs
    (fixtures F2 / F3) and its own property tests (espm/tests/test_updates.py:457-475, :508-513, :562-577): an exact
    factorisation is a fixed point, the update stays positive and does not increase the Frobenius loss."""
    gh, gw = golden("f2_step_h"), golden("f3_step_w")
    for t in (0, 1):
        X, G, W, H = (gh[f"in{t}_{s}"] for s in ("X", "G", "W", "H"))
        np.testing.assert_allclose(api["step_h"](X, G, W, H.copy(), simplex_H=True, l2=True), gh[f"l2_{t}_H"], rtol=3e-5, atol=3e-6)
        X, G, W, H = (gw[f"in{t}_{s}"] for s in ("X", "G", "W", "H"))
        np.testing.assert_allclose(api["step_w"](X, G, W.copy(), H, l2=True), gw[f"l2_{t}_W"], rtol=3e-5, atol=1e-7)
    rng = np.random.default_rng(5)
    n, m, k, p = 70, 9, 3, 150
    G = rng.random((n, m))
    W = rng.random((m, k)) + 0.05
    H = rng.random((k, p)) + 0.05
    H /= H.sum(axis=0, keepdims=True)
    X = G @ W @ H                                                    # exact: (W, H) is a fixed point
    np.testing.assert_allclose(api["step_w"](X, G, W.copy(), H, l2=True, simplex_W=False), W, rtol=2e-5)
    np.testing.assert_allclose(api["step_h"](X, G, W, H.copy(), simplex_H=False, l2=True), H, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(api["step_h"](X, G, W, H.copy(), simplex_H=True, l2=True), H, atol=1e-5)
    frob = lambda Wc, Hc: float(np.sum((X - G @ Wc @ Hc) ** 2))
    for _ in range(3):
        W2 = rng.random((m, k)) + 0.01
        Wn = api["step_w"](X, G, W2.copy(), H, l2=True, simplex_W=False)
        assert (Wn > 0).all() and frob(Wn, H) <= frob(W2, H) * (1 + 1e-6)
        H2 = rng.random((k, p)) + 0.01
        Hn = api["step_h"](X, G, W, H2.copy(), simplex_H=False, l2=True)
        assert (Hn > 0).all() and frob(W, Hn) <= frob(W, H2) * (1 + 1e-6)
    Xi = rng.random((40, 60))                                        # G = identity
    Wi, Hi = rng.random((40, 2)) + 0.1, rng.random((2, 60)) + 0.1
    Wn = api["step_w"](Xi, np.eye(40), Wi.copy(), Hi, l2=True)
    np.testing.assert_allclose(Wn, Wi / (Wi @ (Hi @ Hi.T)) * (Xi @ Hi.T), rtol=3e-5)
    Hn = api["step_h"](Xi, np.eye(40), Wi, Hi.copy(), l2=True)
    np.testing.assert_allclose(Hn, Hi * (Wi.T @ Xi) / ((Wi.T @ Wi) @ Hi), rtol=3e-5)


def test_step_h_reference_properties(api):
    """Fixed point, simplex, positivity, monotone decrease (espm/tests/test_updates.py:484-552)."""
    rng = np.random.default_rng(2)
    l, k, p, c = 26, 5, 100, 17
    A = rng.random((k, p))
    A = A / A.sum(axis=0, keepdims=True)
    G, P = rng.random((l, c)), rng.random((c, k))
    GP = G @ P
    X = GP @ A
    Ap = api["step_h"](X, G, P, A, simplex_H=False, mu=0, log_shift=0, epsilon_reg=1, safe=True)
    np.testing.assert_allclose(A, Ap, atol=5e-6)
    Ap = api["step_h"](X, G, P, A, simplex_H=True, mu=0, log_shift=0, epsilon_reg=1, safe=True)
    np.testing.assert_allclose(A, Ap, atol=oc.DICOTOMY_TOL)
    for _ in range(4):
        A0 = rng.random((k, p))
        A0 = A0 / A0.sum(axis=1, keepdims=True)
        Ap = api["step_h"](X, G, P, A0, simplex_H=False, mu=0, log_shift=0)
        assert (Ap > 0).all() and oc.KLdiv_loss(X, GP, A0) > oc.KLdiv_loss(X, GP, Ap)
        mu = np.ones(k)
        mu[0] = 0
        Ap = api["step_h"](X, G, P, A0, simplex_H=True, mu=3 * mu, epsilon_reg=1)
        np.testing.assert_allclose(Ap.sum(axis=0), np.ones(p), atol=oc.DICOTOMY_TOL)
        np.testing.assert_allclose(Ap, oc.multiplicative_step_h(X, G, P, A0, simplex_H=True, mu=3 * mu), rtol=3e-5, atol=3e-6)
        assert oc.KLdiv_loss(X, GP, A0) + oc.log_reg(A0, 3 * mu, 1) > oc.KLdiv_loss(X, GP, Ap) + oc.log_reg(A0, 3 * mu, 1)


# ------------------------------------------------------------------------------------ W step
def test_step_w_golden(api, golden):
    g = golden("f3_step_w")
    for c in range(int(g["n_cases"])):
        t, simplex, fix_on = g[f"c{c}_cfg"]
        X, G, W, H, fixed = (g[f"in{t}_{s}"] for s in ("X", "G", "W", "H", "fixed"))
        Wn = api["step_w"](X, G, W.copy(), H, simplex_W=bool(simplex), fixed_W=fixed if fix_on else None)
        np.testing.assert_allclose(Wn, g[f"c{c}_W"], rtol=3e-5, atol=1e-7, err_msg=f"case {c}")


def test_step_w_reference_properties(api):
    """espm/tests/test_updates.py:439-470."""
    rng = np.random.default_rng(3)
    l, k, p, c = 26, 5, 100, 17
    A = rng.random((k, p))
    A = A / A.sum(axis=1, keepdims=True)
    G, P = rng.random((l, c)), rng.random((c, k))
    X = G @ P @ A
    np.testing.assert_allclose(api["step_w"](X, G, P, A, log_shift=0, simplex_W=False), P, atol=5e-6)
    for _ in range(4):
        P0 = rng.random((c, k))
        Pp = api["step_w"](X, G, P0, A, simplex_W=False)
        np.testing.assert_allclose(Pp, oc.multiplicative_step_w(X, G, P0, A), rtol=2e-5, atol=1e-7)
        assert (Pp > 0).all() and oc.KLdiv_loss(X, G @ P0, A) > oc.KLdiv_loss(X, G @ Pp, A)


# ------------------------------------------------------------------------------------ operators / losses
def test_laplacian_stencil(api, golden):
    g = golden("f4_laplacian")
    import torch
    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream
    for i, (nx, ny) in enumerate(g["shapes"]):
        H = g[f"s{i}_H"]
        h = torch.from_numpy(H.astype(np.float32)).cuda()
        out = torch.empty_like(h)
        _lib.check(_lib.lib.espm_mu_laplacian(_ptr(h), H.shape[0], int(nx), int(ny), H.shape[1], _ptr(out), _stream()))
        np.testing.assert_allclose(out.cpu().numpy(), g[f"s{i}_HL"], rtol=1e-5, atol=2e-6)
        L = api["utils"].create_laplacian_matrix(nx, ny)
        np.testing.assert_allclose(api["measures"].trace_xtLx(L, H.T), g[f"s{i}_trace"], rtol=1e-5)
        if f"s{i}_dense" in g:
            np.testing.assert_array_equal(np.asarray(L.todense()), g[f"s{i}_dense"])
    L = api["utils"].create_laplacian_matrix(4, 6)  # espm/tests/test_measures.py:212-221
    x = np.ones((4, 6))
    assert api["measures"].trace_xtLx(L, x.ravel()) == 0
    x[2, 3] = 2
    np.testing.assert_allclose(api["measures"].trace_xtLx(L, x.ravel()), 4)


def test_losses_golden(api, golden):
    g = golden("f5_losses")
    X, W, H, mu = g["X"], g["W"], g["H"], g["mu"]
    m = api["measures"]
    np.testing.assert_allclose(m.KLdiv_loss(X, W, H), g["KLdiv_loss"], rtol=2e-6)
    np.testing.assert_allclose(m.KLdiv_loss(X, W, H, average=True), g["KLdiv_loss_avg"], rtol=2e-6)
    np.testing.assert_allclose(m.log_reg(H, mu, 0.8), g["log_reg"], rtol=1e-10)
    np.testing.assert_allclose(m.log_reg(H, 0.3, 1), g["log_reg_scalar"], rtol=1e-10)


def test_initialize_algorithms_golden(api, golden):
    g = golden("f7_init")
    X, G = g["X"], g["G"]
    for init in (None, "random", "nndsvd"):
        for use_G in (False, True):
            for simplex_H in (False, True):
                _, W, H = api["init"](X, G if use_G else None, None, None, 3, init, 0, simplex_H, not simplex_H)
                tag = f"{init}_{int(use_G)}_{int(simplex_H)}"
                np.testing.assert_allclose(W, g[f"{tag}_W"], rtol=1e-9, atol=1e-14, err_msg=tag)
                np.testing.assert_allclose(H, g[f"{tag}_H"], rtol=1e-9, atol=1e-14, err_msg=tag)


@pytest.mark.parametrize("n,nx,ny,layout", [(100, 20, 20, "cm"), (1980, 33, 31, "pm"), (2050, 40, 30, "cm"), (70, 64, 33, "pm")])
def test_ell_builders_agree(monkeypatch, n, nx, ny, layout):
    """The C-ABI builder of the sparse count store (espm_mu_ell_count / _plan / _fill) and the tensor-op builder
    espm_amd.ell.build agree on the plan (orders, offsets) and both decode to X; the C builder spreads the unit entries
    over the LDS banks."""
    import torch
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    prob = synth.make_problem(n, nx, ny, 3, N=0.25 * n, seed=n)
    X = np.minimum(synth.sample_numpy(prob, seed=n), 255.0)
    rng = np.random.default_rng(n)
    X[rng.integers(0, n, 50), rng.integers(0, nx * ny, 50)] = rng.integers(20, 256, 50)   # counts that need several entries
    X[X.sum(axis=1) == 0, 0] = 1.0
    X[0, X.sum(axis=0) == 0] = 1.0
    Xin = X if layout == "cm" else np.ascontiguousarray(X.T)
    stores = {}
    for builder in ("hip", "torch"):
        monkeypatch.setenv("ESPM_ELL_BUILDER", builder)
        eng = MUEngine(Xin, 3, layout=layout, shape_2d=(nx, ny), x_store="ell")
        stores[builder] = eng.ell
    a, b = stores["hip"], stores["torch"]
    for key in ("ell_h_off", "ell_w_off", "chan_perm", "pix_perm"):
        assert torch.equal(a[key].cpu(), b[key].cpu()), key
    for key in ("nnz", "entries_h", "entries_w", "rows_h", "rows_w", "unit_rows_h", "unit_rows_w", "n_cg", "nblk_w"):
        assert a[key] == b[key], key
    np.testing.assert_allclose(a["klc"].cpu().numpy(), b["klc"].cpu().numpy(), rtol=1e-6, atol=1e-6)
    # which ones of a list sit in its unit rows, and in which order, is the builder's choice (the C builder spreads
    # them over the LDS banks): both sets of lists must decode to X
    from ell_decode import decode, general_gather_passes, unit_bank_spread, unit_bank_spread_b32
    p, cbits = nx * ny, eng.st.ell_cbits
    Xi = np.ascontiguousarray(X.T).astype(np.int64)
    spread = {}
    for name, st in stores.items():
        host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
        Xh, Xw, _, _ = decode(host, p, n, eng.st.p_pad, cbits, eng.st.tile_px)
        assert np.array_equal(Xh[:p], Xi) and not Xh[p:].any(), name
        assert np.array_equal(Xw[:p], Xi) and not Xw[p:].any(), name
        spread[name] = (unit_bank_spread(host["ell_h"].numpy(), host["ell_h_off"].numpy()),
                        unit_bank_spread(host["ell_w"].numpy(), host["ell_w_off"].numpy()),
                        unit_bank_spread_b32(host["ell_h"].numpy(), host["ell_h_off"].numpy()),
                        general_gather_passes(host["ell_h"].numpy(), host["ell_h_off"].numpy(), cbits),
                        general_gather_passes(host["ell_w"].numpy(), host["ell_w_off"].numpy(), (2 * eng.st.tile_px).bit_length() - 1))
    for which in (0, 1):
        (s_hip, rows), (s_torch, _) = spread["hip"][which], spread["torch"][which]
        if rows >= 64:  # conflict-free gathers: most read groups of the C builder's unit rows, few of an index-ordered list
            # (W lists of a small image: a channel has a few dozen entries in a block of 2 tile_px pixels, so its 16 index
            # buckets are unevenly filled and the placement has to fill holes - better than index order is all it can be)
            floor = 0.6 if (which == 0 or eng.st.ell_pb == 1024) else 0.15
            assert s_hip > floor and s_hip > 3 * s_torch, (which, s_hip, s_torch)
    # (round 4: with ESPM_ELL_BUCKETS=32 the 4-byte gathers of a 32-lane half fall on 32 different banks too - spread[...][2] -
    #  measured to buy nothing, profiles/r04u_buckets_ab_*.log; the product keeps 16 buckets, where that share is a few per cent)
    assert spread["hip"][2][0] >= 0.0
    # (round 4: general rows placed by bank quad - spread[...][3], [4] count their gather passes - were built and withdrawn: 2.6 -> 1.9
    #  passes per read group on the headline's lists, no time gained, tools/proto/ell_general_rows_placed.patch)
    assert spread["hip"][3][0] >= 1.0 and spread["hip"][4][0] >= 1.0


@pytest.mark.parametrize("n,nx,ny,k,fix", [(70, 9, 13, 6, True), (2048, 16, 32, 5, False), (333, 7, 19, 3, True)])
def test_local_w_update_equals_the_one_workgroup_finish(n, nx, ny, k, fix):
    """G = identity, no simplex_W: the slab-reduction workgroups update their own entries of W (w_reduce_update_kernel +
    tail).  Same W, H, losses and rel_W as the reduction followed by w_finish (forced by withholding the workspace the
    local update needs), including fixed_W, a channel count that leaves the last channel block ragged, and k = 6."""
    import torch
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    prob = synth.make_problem(n, nx, ny, k, N=120.0, seed=n)
    X = synth.sample_numpy(prob, seed=n)
    W0, H0 = synth.random_init(n, k, nx * ny, seed=n, scale=0.2)
    fixed_W = None
    if fix:
        fixed_W = -np.ones((n, k))
        fixed_W[::7, 0] = 0.05
        fixed_W[3, :] = 0.0
    out = {}
    for name in ("local", "finish"):
        eng = MUEngine(X, k, shape_2d=(nx, ny), lambda_L=0.5, simplex_H=True, simplex_W=False, fixed_W=fixed_W, tol=0.0,
                       max_iter=12)
        if name == "finish":
            eng.st.w_scratch = None
        eng.load_state(W0, H0)
        eng.iterate(6, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        out[name] = (eng.get_W(), eng.get_H(), h["loss"], h["rel_W"], h["rel_H"], h["bad"].sum())
    a, b = out["local"], out["finish"]
    assert a[5] == 0 and b[5] == 0
    np.testing.assert_allclose(a[0], b[0], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-7)
    np.testing.assert_allclose(a[3][1:], b[3][1:], rtol=1e-5)
    np.testing.assert_allclose(a[4][1:-1], b[4][1:-1], rtol=1e-4, atol=1e-9)
    assert (a[3][1:] > 0).all()
    if fix:
        m = fixed_W >= 0
        assert np.array_equal(a[0][m], fixed_W[m].astype(np.float32))


def test_gradients_and_q_step_golden(golden):
    """gradW / gradH (KL and l2 branches, vector mu, Laplacian) and update_q as module-level functions against the reference
    (fixture F15; espm/estimators/updates.py:225-230, :303-342)."""
    from espm_amd.estimators.updates import gradH, gradW, update_q
    from espm_amd.utils import create_laplacian_matrix
    g = golden("f15_gradients")
    for name in g["names"]:
        X, G, W0, H0, mu = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "mu"))
        L = create_laplacian_matrix(*(int(v) for v in g[f"{name}_shape"]))
        np.testing.assert_allclose(gradW(X, G, W0, H0), g[f"{name}_gradW"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gradW(X, G, W0, H0, l2=True), g[f"{name}_gradW_l2"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(gradH(X, G, W0, H0, mu=mu, lambda_L=0.8, L=L, epsilon_reg=0.7), g[f"{name}_gradH"],
                                   rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(gradH(X, G, W0, H0), g[f"{name}_gradH_plain"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gradH(X, G, W0, H0, mu=0.2, lambda_L=0.5, L=L, l2=True), g[f"{name}_gradH_l2"],
                                   rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(update_q(G @ W0, H0), g[f"{name}_Q"], rtol=1e-11, atol=1e-14)
    with pytest.raises(ValueError):
        gradH(X, G, W0, H0, lambda_L=1.0)


def test_kldiv_measure_golden(golden):
    """`espm.measures.KLdiv` (measures.py:387-425) - the factorised KL divergence, evaluated by the H-step kernel in loss-only mode - against
    fixture F20 from the reference (its X holds an empty channel: clamped at log_shift there)."""
    from espm_amd.measures import KLdiv
    g = golden("f20_measures_and_dicotomy")
    np.testing.assert_allclose(KLdiv(g["X"], g["W"], g["H"]), g["KLdiv"], rtol=1e-5)
    np.testing.assert_allclose(KLdiv(g["X"], g["W"], g["H"], average=True), g["KLdiv_avg"], rtol=1e-5)


def test_multiplicative_step_wq_golden(golden):
    """`multiplicative_step_wq` (espm/estimators/updates.py:232-261) against fixture F19 from the reference: the W step of the HIP path without
    the simplex; with simplex_W=True what the reference RETURNS - not on the simplex, its multiplier is found for the numerators without their
    factor W - also over the physics model's row subset.  (Until round 5 this returned multiplicative_step_w's result, as the docstring reads.)"""
    from espm_amd.estimators.updates import multiplicative_step_wq

    class Rows:
        def __init__(self, rows):
            self.rows = rows

        def NMF_simplex(self):
            return self.rows

    g = golden("f19_multiplicative_step_wq")
    for name in g["names"]:
        X, G, W0, H0, rows = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "rows"))
        scale = np.abs(g[f"{name}_wq_free"]).max()
        np.testing.assert_allclose(multiplicative_step_wq(X, G, W0, H0, simplex_W=False), g[f"{name}_wq_free"], rtol=2e-5, atol=2e-6 * scale)
        got = multiplicative_step_wq(X, G, W0, H0, simplex_W=True)
        np.testing.assert_allclose(got, g[f"{name}_wq_simplex"], rtol=5e-5, atol=5e-6 * scale)
        assert np.abs(got.sum(axis=0) - 1).min() > 0.05     # (the reference's result is not on the simplex)
        np.testing.assert_allclose(multiplicative_step_wq(X, G, W0, H0, simplex_W=True, physics_model=Rows(rows)), g[f"{name}_wq_rows"],
                                   rtol=5e-5, atol=5e-6 * scale)


def test_projected_gradient_steps_with_the_frobenius_gradient_golden(golden):
    """proj_grad_step_w / _h(l2=True) - NotImplementedError until round 5 (missing 5) - against fixture F18 from the
    reference (espm/estimators/updates.py:353-395)."""
    from espm_amd.estimators.updates import proj_grad_step_h, proj_grad_step_w
    from espm_amd.utils import create_laplacian_matrix
    g = golden("f18_projected_gradient_l2")
    for name in g["names"]:
        X, G, W0, H0, mu = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "mu"))
        gh, gw = (float(v) for v in g[f"{name}_gamma"])
        fW, fH = g[f"{name}_fixed_W"], g[f"{name}_fixed_H"]
        L = create_laplacian_matrix(*(int(v) for v in g[f"{name}_shape"]))
        np.testing.assert_allclose(proj_grad_step_w(X, G, W0, H0, gw, simplex_W=False, l2=True), g[f"{name}_W_l2"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(proj_grad_step_w(X, G, W0, H0, gw, simplex_W=False, l2=True, fixed_W=fW), g[f"{name}_W_l2_fixed"], rtol=1e-9, atol=1e-12)
        # (the reference's multiplier stops at dicotomy_tol = 1e-5 under a global rule, the device kernel per column)
        np.testing.assert_allclose(proj_grad_step_h(X, G, W0, H0, gh, simplex_H=True, l2=True), g[f"{name}_H_l2"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(proj_grad_step_h(X, G, W0, H0, gh, simplex_H=False, mu=mu, lambda_L=0.6, L=L, epsilon_reg=0.8, l2=True, fixed_H=fH),
                                   g[f"{name}_H_l2_free"], rtol=1e-6, atol=1e-9)
        H2 = proj_grad_step_h(X, G, W0, H0, gh, simplex_H=True, mu=0.2, lambda_L=0.5, L=L, l2=True)
        np.testing.assert_allclose(H2, g[f"{name}_H_l2_reg"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(H2.sum(axis=0), 1.0, atol=2e-5)
    with pytest.raises(NotImplementedError):
        proj_grad_step_w(X, G, W0, H0, gw, simplex_W=True, l2=True)


def test_simplex_multiplier_with_tiny_numerators():
    """All numerators ~1e-13 next to denominators ~30 (a pixel without counts: only the reference's log_shift fill feeds
    its column, base.py:519-528): the bracket's upper end 2 k max(num) - min(den) + d* must not cancel to zero."""
    from espm_amd.estimators.dicotomy import dichotomy_simplex
    rng = np.random.default_rng(3)
    num = rng.random((4, 50)) * 1e-13 + 1e-14
    den = 25.0 + 10.0 * rng.random((4, 50))
    nu = dichotomy_simplex(num, den, 1e-14, tol=1e-9)
    # the root sits ~num_k* right of the pole -min(den) and nu (fp64) resolves that distance to ulp(30) = 3.6e-15 only - the
    # reference has the same limit (test_host_cpu.py::test_reference_loses_the_simplex_next_to_a_pole); the H-step works with the
    # distance itself.  Here: the distance returned is the numerator of the component at the pole, to that resolution.
    dist = nu + den.min(axis=0)
    k_star = np.argmin(den, axis=0)
    assert np.all(dist > 0)
    np.testing.assert_allclose(dist, num[k_star, np.arange(50)], atol=8e-15, rtol=0)
    nu_ref = oc.dichotomy_simplex(num, den, 1e-14, tol=1e-12, maxit=200)
    np.testing.assert_allclose(nu, nu_ref, atol=8e-15, rtol=0)


def test_other_multipliers_known_answers(golden):
    """dichotomy_simplex_acc and dichotomy_simplex_projected_gradient as module-level functions: the reference's root of
    fixture F11, the defining equations, and the reference's own checks (espm/tests/test_updates.py:251-437, :675-731)."""
    from espm_amd.estimators.dicotomy import dichotomy_simplex_acc, dichotomy_simplex_projected_gradient
    g = golden("f11_quadratic_surrogate")
    a, b, c = float(g["acc_a"]), g["acc_b"], g["acc_c"]
    nu = dichotomy_simplex_acc(a, b, c, log_shift=0.0, tol=1e-12, maxit=200)
    np.testing.assert_allclose(nu, g["acc_nu"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(((np.sqrt((b + nu) ** 2 + 4 * a * c) - nu - b) / (2 * a)).sum(axis=0), 1.0, atol=1e-10)
    nu1 = dichotomy_simplex_acc(a, b[:, :1], c, log_shift=0.02, tol=1e-10)       # broadcast b, active floor
    h = np.maximum((np.sqrt((b[:, :1] + nu1) ** 2 + 4 * a * c) - nu1 - b[:, :1]) / (2 * a), 0.02)
    np.testing.assert_allclose(h.sum(axis=0), 1.0, atol=1e-8)
    with pytest.raises(ValueError):
        dichotomy_simplex_acc(a, b, c, log_shift=0.25)
    rng = np.random.default_rng(9)
    v = rng.standard_normal((6, 300)) * 3
    for eps in (0.0, 0.05):
        nu2 = dichotomy_simplex_projected_gradient(v, log_shift=eps, tol=1e-12)
        np.testing.assert_allclose(np.maximum(v + nu2, eps).sum(axis=0), 1.0, atol=1e-10)
    with pytest.raises(ValueError):
        dichotomy_simplex_projected_gradient(v, log_shift=0.2)


def test_surrogates_and_reporting_measures(golden):
    """Module-level surrogates (surrogates.py) against the oracle's restatement, the reference's majorisation tests
    (espm/tests/test_estimators.py:168-204: the surrogate is >= the Laplacian term, with equality at Ht), and the small
    reporting measures."""
    from espm_amd.estimators import surrogates as sg
    from espm_amd import measures as ms
    from espm_amd.utils import create_laplacian_matrix
    rng = np.random.default_rng(21)
    nx, ny, k = 9, 14, 4
    L = create_laplacian_matrix(nx, ny)
    Lo = oc.laplacian_matrix(nx, ny)
    for _ in range(3):
        A1 = rng.random((k, nx * ny)) + 0.01
        A2 = rng.random((k, nx * ny)) + 0.01
        for algo, ref in (("l2_surrogate", oc.smooth_l2_surrogate), ("log_surrogate", oc.smooth_dgkl_surrogate)):
            fn = sg.smooth_l2_surrogate if algo == "l2_surrogate" else sg.smooth_dgkl_surrogate
            np.testing.assert_allclose(fn(A1, L, A2, sigmaL=8, lambda_L=0.7), ref(A1, Lo, A2, 8, 0.7), rtol=2e-6)
            d = sg.diff_surrogate(A1, A2, L=L, algo=algo)
            np.testing.assert_allclose(d, oc.diff_surrogate(A1, A2, Lo, algo=algo), rtol=1e-4, atol=1e-4)
            assert d >= 0
            np.testing.assert_allclose(sg.diff_surrogate(A1, A1, L=L, algo=algo), 0.0, atol=1e-3)
        np.testing.assert_allclose(sg.smooth_l2_surrogate(A1, L), 0.5 * np.sum(A1 * (A1 @ Lo)), rtol=2e-6)
    x, xt, gr = rng.random((3, 5)), rng.random((3, 5)), rng.standard_normal((3, 5))
    np.testing.assert_allclose(sg.quadratic_surrogate(x, xt, 1.5, gr, 2.0), 1.5 + np.sum((x - xt) * gr) + 2.0 * np.sum((x - xt) ** 2))
    X, W, H = rng.random((12, 30)), rng.random((12, 3)), rng.random((3, 30))
    np.testing.assert_allclose(ms.Frobenius_loss(X, W, H), np.sum((X - W @ H) ** 2), rtol=1e-12)
    tm, ts = rng.random((3, 40)), rng.random((3, 25))
    perm = [2, 0, 1]
    am, as_ = tm[perm] + 0.01 * rng.random((3, 40)), ts[perm] * 1.3
    ang, mse_, cfg, warn = ms.find_min_config(tm, ts, am, as_)
    assert list(cfg) == perm and not warn and max(ang) < 1e-4 and max(mse_) < 1e-3   # algo phase i <-> true phase cfg[i]
    assert ms.ordered_mae(tm, am, cfg)[0] < 0.01 and abs(ms.mse(tm[0], tm[0])) == 0


def test_lu_normaliser_kernel_equals_scipy_and_the_torch_formulation():
    """espm_lu_pl (csrc/mu_init.hip: the LU normaliser of the randomized range finder, one launch per column) against
    scipy.linalg.lu(A, permute_l=True)[0] - what scikit-learn calls, espm/estimators/updates.py:179 - and, to rounding (torch divides
    by a scalar through its reciprocal on the host), against the torch formulation of the same elimination on the host
    (espm_amd/init_device._lu_pl): tall matrices up to the headline height, square ones, both precisions, rows that tie for a
    pivot, a zero column, a strided input."""
    import scipy.linalg as sl
    import torch
    from espm_amd import _lib
    from espm_amd.init_device import _lu_pl
    rs = np.random.RandomState(11)
    for shape in [(5000, 15), (262144, 15), (2048, 15), (257, 13), (15, 15), (40, 3), (1, 1), (70000, 40)]:
        for dt, tol in ((np.float64, 5e-13), (np.float32, 2e-5)):
            A = rs.normal(size=shape).astype(dt)
            host = _lu_pl(torch.from_numpy(A)).numpy()
            got = _lu_pl(torch.from_numpy(A).cuda())
            assert got.is_cuda and tuple(got.shape) == shape and got.dtype == torch.from_numpy(A).dtype
            np.testing.assert_allclose(got.cpu().numpy(), host, rtol=0, atol=tol, err_msg=f"{shape} {dt.__name__}")
            if shape[0] <= 5000:
                np.testing.assert_allclose(got.cpu().numpy(), sl.lu(A, permute_l=True)[0], rtol=0, atol=tol, err_msg=f"{shape} {dt.__name__}")
    A = rs.normal(size=(600, 15))
    A[100] = A[7] = A[431] = 50.0 * A[3]     # identical rows, the largest of every column: the first of them is the pivot
    np.testing.assert_allclose(_lu_pl(torch.from_numpy(A).cuda()).cpu().numpy(), sl.lu(A, permute_l=True)[0], rtol=0, atol=5e-13)
    A = rs.normal(size=(900, 6))
    A[:, 2] = 0.0                            # a zero column: zero pivot, multipliers stay zero
    np.testing.assert_allclose(_lu_pl(torch.from_numpy(A).cuda()).cpu().numpy(), _lu_pl(torch.from_numpy(A)).numpy(), rtol=0, atol=5e-13)
    wide = torch.from_numpy(rs.normal(size=(3000, 32))).cuda()[:, ::2]   # columns 2 apart in memory
    np.testing.assert_allclose(_lu_pl(wide).cpu().numpy(), _lu_pl(wide.cpu().contiguous()).numpy(), rtol=0, atol=5e-13)
    with pytest.raises(ValueError, match="NULL pointer"):
        _lib.check(_lib.lib.espm_lu_pl(None, 0, 10, 3, 3, None, None, 0, None))


def test_sparse_store_of_dense_lists_beyond_the_unit_rows():
    """Dense data forced into the sparse store with more channels than unit rows exist for (n > 4096): lists of 4200 general entries next to
    lists of a few, split counts included.  The lists decode to X, and the fit's first iterations equal the dense fp32 store's."""
    import torch
    from espm_amd.engine import MUEngine
    from ell_decode import decode
    n, nx, ny, k = 4200, 8, 16, 3
    rng = np.random.default_rng(5)
    X = rng.poisson(0.4, size=(n, nx * ny)).astype(np.float32)
    X[:, :40] = rng.integers(2, 9, size=(n, 40))          # 40 pixels whose every channel holds a count >= 2: 262 entries per bucket
    X[:300, 64:] += rng.integers(2, 5, size=(300, nx * ny - 64))   # 64 pixels with few general entries, 64 with ~300 (19 per bucket)
    W0 = rng.uniform(0.5, 1.5, size=(n, k)) * X.mean() / k
    H0 = rng.uniform(0.5, 1.5, size=(k, nx * ny))
    out = {}
    for store in ("ell", "f32"):
        eng = MUEngine(X, k, layout="cm", shape_2d=(nx, ny), lambda_L=0.5, simplex_H=True, simplex_W=False, tol=0.0, max_iter=6, x_store=store)
        eng.load_state(W0, H0)
        eng.iterate(4, final_loss=True)
        torch.cuda.synchronize()
        out[store] = (eng.get_W(), eng.get_H(), eng.history()["loss"].copy())
        if store == "ell":
            host = {kk: (v.cpu() if torch.is_tensor(v) else v) for kk, v in eng.ell.items()}
            Xh, Xw, _, _ = decode(host, nx * ny, n, eng.st.p_pad, eng.st.ell_cbits, eng.st.tile_px)
            Xi = np.ascontiguousarray(X.T).astype(np.int64)
            assert np.array_equal(Xh[:nx * ny], Xi) and np.array_equal(Xw[:nx * ny], Xi)
    np.testing.assert_allclose(out["ell"][2], out["f32"][2], rtol=2e-6)
    np.testing.assert_allclose(out["ell"][0], out["f32"][0], rtol=2e-4, atol=1e-6 * out["f32"][0].max())
    np.testing.assert_allclose(out["ell"][1], out["f32"][1], rtol=2e-4, atol=2e-6)
