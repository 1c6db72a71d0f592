"""Pins the numpy oracle (oracle/mu_oracle.py) to golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from oracle import mu_oracle as oc

RT = dict(rtol=1e-12, atol=1e-14)


def test_multiplicative_step_wq_golden(golden):
    """F19: `multiplicative_step_wq` as the reference runs it (updates.py:232-261) - without the simplex the same as
    `multiplicative_step_w`, with it NOT on the simplex (the multiplier is found for the numerators without their factor W)."""
    g = golden("f19_multiplicative_step_wq")

    for name in g["names"]:
        X, G, W0, H0, rows = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "rows"))
        np.testing.assert_allclose(oc.multiplicative_step_wq(X, G, W0, H0, simplex_W=True), g[f"{name}_wq_simplex"], rtol=1e-9)
        np.testing.assert_allclose(oc.multiplicative_step_wq(X, G, W0, H0, simplex_W=False), g[f"{name}_wq_free"], rtol=1e-10)
        np.testing.assert_allclose(oc.multiplicative_step_wq(X, G, W0, H0, simplex_W=True, rows=rows), g[f"{name}_wq_rows"], rtol=1e-9)
        np.testing.assert_allclose(g[f"{name}_wq_free"], g[f"{name}_w_free"], rtol=1e-10)           # what its docstring promises ...
        assert np.abs(g[f"{name}_wq_simplex"].sum(axis=0) - 1).min() > 0.05                            # ... and what it does not
        np.testing.assert_allclose(g[f"{name}_w_simplex"].sum(axis=0), 1.0, atol=2e-5)


def test_f1_dichotomy_known_answers(golden):
    g = golden("f1_dichotomy")
    for i in range(int(g["n_kat"])):
        nu = oc.dichotomy_simplex(g[f"kat{i}_num"], g[f"kat{i}_den"], float(g[f"kat{i}_eps"]), tol=float(g[f"kat{i}_tol"]))
        np.testing.assert_allclose(nu, g[f"kat{i}_nu"], **RT)
    # espm/tests/test_updates.py:146-151: the answer is nu = 3
    assert abs(g["kat2_nu"][0] - 3) < 2e-8


def test_f1_dichotomy_random_scales(golden):
    g = golden("f1_dichotomy")
    for c in range(int(g["n_rnd"])):
        num, den, eps = g[f"rnd{c}_num"], g[f"rnd{c}_den"], float(g[f"rnd{c}_eps"])
        for tol, tag in ((oc.DICOTOMY_TOL, "t5"), (0.0, "t0")):
            nu = oc.dichotomy_simplex(num, den, eps, tol=tol, maxit=100)
            np.testing.assert_allclose(nu, g[f"rnd{c}_{tag}_nu"], rtol=1e-12, atol=0)


def test_dichotomy_errors():
    rng = np.random.default_rng(0)
    with pytest.raises(ValueError):  # k * log_shift >= 1, espm/tests/test_updates.py:160-167
        oc.dichotomy_simplex(rng.random((3, 10)), rng.random((3, 10)), 0.5)
    with pytest.raises(AssertionError):
        oc.dichotomy_simplex(np.zeros((3, 4)), np.ones((3, 4)), 0.0)


def _step_h_inputs(g, t):
    return g[f"in{t}_X"], g[f"in{t}_G"], g[f"in{t}_W"], g[f"in{t}_H"], g[f"in{t}_fixed"]


def test_f2_step_h_grid(golden):
    g = golden("f2_step_h")
    nx, ny = g["shape_2d"]
    L = oc.laplacian_matrix(nx, ny)
    for c in range(int(g["n_cases"])):
        t, simplex, lam, mu_on, fix_on = g[f"c{c}_cfg"]
        X, G, W, H, fixed = _step_h_inputs(g, int(t))
        Hn = oc.multiplicative_step_h(X, G, W, H.copy(), simplex_H=bool(simplex), mu=g["mu_vec"] if mu_on else 0,
                                      epsilon_reg=float(g["epsilon_reg"]), lambda_L=float(lam), L=L,
                                      fixed_H=fixed if fix_on else None)
        np.testing.assert_allclose(Hn, g[f"c{c}_H"], rtol=1e-11, atol=1e-15, err_msg=f"case {c}")


def test_f2_step_h_special(golden):
    g = golden("f2_step_h")
    nx, ny = g["shape_2d"]
    L = oc.laplacian_matrix(nx, ny)
    for t in (0, 1):
        X, G, W, H, _ = _step_h_inputs(g, t)
        np.testing.assert_allclose(oc.multiplicative_step_h(X, G, W, H.copy(), simplex_H=True, l2=True),
                                   g[f"l2_{t}_H"], rtol=1e-11)
        np.testing.assert_allclose(oc.multiplicative_step_h(X, G, W, H.copy(), simplex_H=True, mu=0.4, lambda_L=0.5,
                                                            L=L, sigmaL=11.0), g[f"smu_{t}_H"], rtol=1e-11)
        np.testing.assert_allclose(oc.multiplicative_step_h(X, G, W, H.copy(), simplex_H=True, lambda_L=1.5,
                                                            L=oc.identity_L(nx * ny)), g[f"lid_{t}_H"], rtol=1e-11)
    with pytest.raises(ValueError):
        oc.multiplicative_step_h(X, G, W, H, lambda_L=1.0, L=None)


def test_f3_step_w(golden):
    g = golden("f3_step_w")
    for c in range(int(g["n_cases"])):
        t, simplex, fix_on = g[f"c{c}_cfg"]
        X, G, W, H, fixed = (g[f"in{t}_{s}"] for s in ("X", "G", "W", "H", "fixed"))
        Wn = oc.multiplicative_step_w(X, G, W.copy(), H, simplex_W=bool(simplex), fixed_W=fixed if fix_on else None)
        np.testing.assert_allclose(Wn, g[f"c{c}_W"], rtol=1e-11, atol=1e-15, err_msg=f"case {c}")
    for t in (0, 1):
        X, G, W, H = (g[f"in{t}_{s}"] for s in ("X", "G", "W", "H"))
        np.testing.assert_allclose(oc.multiplicative_step_w(X, G, W.copy(), H, l2=True), g[f"l2_{t}_W"], rtol=1e-11)


def test_f4_laplacian(golden):
    g = golden("f4_laplacian")
    for i, (nx, ny) in enumerate(g["shapes"]):
        L = oc.laplacian_matrix(nx, ny)
        H = g[f"s{i}_H"]
        np.testing.assert_allclose(H @ L, g[f"s{i}_HL"], **RT)
        np.testing.assert_allclose(oc.laplacian_apply(H, nx, ny), g[f"s{i}_HL"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(oc.trace_xtLx(L, H.T), g[f"s{i}_trace"], rtol=1e-12)
        if f"s{i}_dense" in g:
            np.testing.assert_array_equal(np.asarray(L.todense()), g[f"s{i}_dense"])
    # espm/tests/test_measures.py:212-221: 0 for a constant map, 4 for a unit bump
    L = oc.laplacian_matrix(4, 6)
    x = np.ones((4, 6))
    assert oc.trace_xtLx(L, x.ravel()) == 0
    x[2, 3] = 2
    np.testing.assert_allclose(oc.trace_xtLx(L, x.ravel()), 4)


def test_f5_losses(golden):
    g = golden("f5_losses")
    X, W, H, mu = g["X"], g["W"], g["H"], g["mu"]
    np.testing.assert_allclose(oc.KLdiv_loss(X, W, H), g["KLdiv_loss"], rtol=1e-13)
    np.testing.assert_allclose(oc.KLdiv_loss(X, W, H, average=True), g["KLdiv_loss_avg"], rtol=1e-13)
    np.testing.assert_allclose(oc.Frobenius_loss(X, W, H), g["Frobenius_loss"], rtol=1e-13)
    np.testing.assert_allclose(oc.log_reg(H, mu, 0.8), g["log_reg"], rtol=1e-13)
    np.testing.assert_allclose(oc.log_reg(H, 0.3, 1), g["log_reg_scalar"], rtol=1e-13)
    X_ = oc.remove_zeros_lines(X, oc.LOG_SHIFT)
    np.testing.assert_array_equal(X_, g["X_"])
    np.testing.assert_allclose(oc.const_KL(X_), g["const_KL"], rtol=1e-13)
    nx, ny = g["shape_2d"]
    L = oc.laplacian_matrix(nx, ny)
    G = np.eye(X.shape[0])
    for avg in (True, False):
        tot, det = oc.smooth_nmf_loss(X_, G, W, H, L, mu, 0.8, 1.5, average=avg)
        np.testing.assert_allclose(tot, g[f"loss_avg{int(avg)}"], rtol=1e-12)
        np.testing.assert_allclose(det, g[f"detailed_avg{int(avg)}"], rtol=1e-11)


def test_f6_trajectories(golden):
    g = golden("f6_trajectories")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        G = g.get(f"{name}_G")
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        for mode, extra in (("free", dict(tol=0, no_stop_criterion=True, max_iter=50)),
                            ("stop", dict(tol=1e-3, max_iter=200))):
            r = oc.fit(g[f"{name}_X"], c["k"], G=G, W=g[f"{name}_W0"].copy(), H=g[f"{name}_H0"].copy(),
                       shape_2d=shape, record_at=(1, 2, 5, 50), **c["kw"], **extra)
            pre = f"{name}_{mode}"
            assert r["n_iter"] == int(g[f"{pre}_n_iter"]), pre
            np.testing.assert_allclose(r["losses"], g[f"{pre}_losses"], rtol=1e-9, err_msg=pre)
            np.testing.assert_allclose(r["detailed_losses"], g[f"{pre}_detailed"], rtol=1e-9, atol=1e-18, err_msg=pre)
            np.testing.assert_allclose(r["rel"], g[f"{pre}_rel"], rtol=1e-7, atol=1e-12, err_msg=pre)
            np.testing.assert_allclose(r["W"], g[f"{pre}_W"], rtol=1e-8, atol=1e-14, err_msg=pre)
            np.testing.assert_allclose(r["H"], g[f"{pre}_H"], rtol=1e-8, atol=1e-14, err_msg=pre)
            np.testing.assert_allclose(r["GW"], g[f"{pre}_GW"], rtol=1e-8, atol=1e-14, err_msg=pre)
            np.testing.assert_allclose(r["reconstruction_err"], g[f"{pre}_recon"], rtol=1e-9)
            if mode == "free":
                for t, (Wt, Ht) in r["snapshots"].items():
                    np.testing.assert_allclose(Wt, g[f"{pre}_W{t}"], rtol=1e-8, atol=1e-14)
                    np.testing.assert_allclose(Ht, g[f"{pre}_H{t}"], rtol=1e-8, atol=1e-14)


def test_f7_init(golden):
    g = golden("f7_init")
    X, G = g["X"], g["G"]
    for init in (None, "random", "nndsvd"):
        for use_G in (False, True):
            for simplex_H in (False, True):
                _, W, H = oc.initialize_algorithms(X, G if use_G else None, None, None, 3, init, 0, simplex_H,
                                                   not simplex_H)
                tag = f"{init}_{int(use_G)}_{int(simplex_H)}"
                np.testing.assert_allclose(W, g[f"{tag}_W"], rtol=1e-9, atol=1e-14, err_msg=tag)
                np.testing.assert_allclose(H, g[f"{tag}_H"], rtol=1e-9, atol=1e-14, err_msg=tag)
    _, _, H = oc.initialize_algorithms(X, G, g["Wgiven_W0"], None, 3, None, 0, True, False)
    np.testing.assert_allclose(H, g["Wgiven_H"], rtol=1e-9, atol=1e-14)
    _, W, _ = oc.initialize_algorithms(X, G, None, g["Hgiven_H0"], 3, None, 0, True, False)
    np.testing.assert_allclose(W, g["Hgiven_W"], rtol=1e-9, atol=1e-14)


def test_f8_fit_outputs(golden):
    g = golden("f8_api")
    X, W0, H0 = g["hspy_X"], g["hspy_W0"], g["hspy_H0"]
    r = oc.fit(X, 2, W=W0.copy(), H=H0.copy(), simplex_H=True, simplex_W=False, max_iter=3)
    np.testing.assert_allclose(r["H"].T, g["hspy_ret"], rtol=1e-9)
    np.testing.assert_allclose(r["GW"].T, g["hspy_components"], rtol=1e-9)
    r = oc.fit(g["norm_X"], 5, lambda_L=1.0, max_iter=10, init="nndsvd", normalize=True, shape_2d=(8, 4),
               random_state=0, simplex_W=False, simplex_H=True)
    np.testing.assert_allclose(r["norm_factor"], g["norm_factor"], rtol=1e-12)
    np.testing.assert_allclose(r["GW"], g["norm_GP"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(r["H"], g["norm_H"], rtol=1e-7, atol=1e-12)
    r = oc.fit(X, 2, W=W0.copy(), H=H0.copy(), simplex_H=False, simplex_W=False, max_iter=4)
    np.testing.assert_allclose(r["GW"], g["nosimplex_GW"], rtol=1e-8)
    np.testing.assert_allclose(r["H"], g["nosimplex_H"], rtol=1e-8)
    np.testing.assert_allclose(r["reconstruction_err"], g["nosimplex_recon"], rtol=1e-9)


def test_f9_linesearch_and_truth_tracking(golden):
    """The options of SURVEY 8(f) rank 4 that are built: linesearch on the Laplacian surrogate (gamma adapts every
    iteration, smooth_nmf.py:376-381) and true_D / true_H tracking (base.py:301-347)."""
    g = golden("f9_linesearch_truth")
    cfgs = json.loads(str(g["configs"]))
    for name in list(g["names_ls"]) + list(g["names_tm"]):
        c = cfgs[name]
        G = g.get(f"{name}_G")
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        extra = dict(tol=0, no_stop_criterion=True, max_iter=c["iters"])
        if name in g["names_ls"]:
            extra["linesearch"] = True
        else:
            extra.update(true_D=g[f"{name}_true_D"], true_H=g[f"{name}_true_H"])
        r = oc.fit(g[f"{name}_X"], c["k"], G=G, W=g[f"{name}_W0"].copy(), H=g[f"{name}_H0"].copy(), shape_2d=shape,
                   **c["kw"], **extra)
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-9, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["rel"], g[f"{name}_rel"], rtol=1e-7, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=1e-8, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=1e-8, atol=1e-14, err_msg=name)
        if name in g["names_ls"]:
            gam = g[f"{name}_detailed"][:, 3]
            assert len(set(np.round(gam, 9))) > 5, "the fixture must exercise the adaptation"
        else:
            np.testing.assert_allclose(r["angles"], g[f"{name}_angles"], rtol=1e-7, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(r["mse"], g[f"{name}_mse"], rtol=1e-8, atol=1e-16, err_msg=name)
            np.testing.assert_allclose(r["true_losses"], g[f"{name}_true_losses"], rtol=1e-9, err_msg=name)


def test_f10_bregman_variant(golden):
    """use_bregman=True in both step functions and algo="bmd" trajectories (updates.py:40-48, :120-125)."""
    g = golden("f10_bregman")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        X, W0, H0 = g[f"{name}_X"], g[f"{name}_W0"], g[f"{name}_H0"]
        G = np.eye(c["n"])
        L = oc.laplacian_matrix(*shape)
        kw = dict(c["kw"])
        ls = kw.pop("linesearch", False)
        Hs = oc.multiplicative_step_h(X, G, W0, H0.copy(), simplex_H=kw["simplex_H"], mu=kw["mu"], lambda_L=kw["lambda_L"], L=L,
                                      sigmaL=8, use_bregman=True)
        np.testing.assert_allclose(Hs, g[f"{name}_step_H"], rtol=1e-9, atol=1e-14, err_msg=name)
        Ws = oc.multiplicative_step_w(X, G, W0.copy(), H0, use_bregman=True)
        np.testing.assert_allclose(Ws, g[f"{name}_step_W"], rtol=1e-10, err_msg=name)
        r = oc.fit(X, c["k"], W=W0.copy(), H=H0.copy(), shape_2d=shape, algo="bmd", linesearch=ls, tol=0, no_stop_criterion=True,
                   max_iter=c["iters"], **kw)
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-9, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=1e-8, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=1e-8, atol=1e-14, err_msg=name)


def test_f11_quadratic_surrogate(golden):
    """algo="l2_surrogate": multiplicative_step_hq, its root finder dichotomy_simplex_acc and whole fits
    (updates.py:263-315, dicotomy.py:57-82, surrogates.py:6-58)."""
    g = golden("f11_quadratic_surrogate")
    nu = oc.dichotomy_simplex_acc(float(g["acc_a"]), g["acc_b"].copy(), g["acc_c"].copy(), log_shift=0.0, tol=1e-12, maxit=200)
    np.testing.assert_allclose(nu, g["acc_nu"], rtol=1e-9, atol=1e-10)
    a, b, c = float(g["acc_a"]), g["acc_b"], g["acc_c"]
    np.testing.assert_allclose(((np.sqrt((b + nu) ** 2 + 4 * a * c) - nu - b) / (2 * a)).sum(axis=0), 1.0, atol=1e-9)
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        X, W0, H0 = g[f"{name}_X"], g[f"{name}_W0"], g[f"{name}_H0"]
        G = g.get(f"{name}_G")
        Gd = np.eye(c["n"]) if G is None else G
        kw = dict(c["kw"])
        ls = kw.pop("linesearch", False)
        Hs = oc.multiplicative_step_hq(X, Gd, W0, H0.copy(), simplex_H=kw["simplex_H"], lambda_L=kw["lambda_L"],
                                       L=oc.laplacian_matrix(*shape), sigmaL=8)
        np.testing.assert_allclose(Hs, g[f"{name}_step_H"], rtol=1e-9, atol=1e-14, err_msg=name)
        r = oc.fit(X, c["k"], G=G, W=W0.copy(), H=H0.copy(), shape_2d=shape, algo="l2_surrogate", linesearch=ls, tol=0,
                   no_stop_criterion=True, max_iter=c["iters"], **kw)
        # (single steps agree to 1e-9; over a fit the GLOBAL 1e-5 stop rule of the bisection turns rounding into a sweep more
        #  or less, i.e. into differences of the multiplier of up to its tolerance)
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-7, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-6, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=2e-5, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=2e-5, atol=1e-8, err_msg=name)


def test_f14_frobenius_fit(golden):
    """l2=True inside a fit (kept only with algo="l2_surrogate", smooth_nmf.py:223-237): Frobenius W step
    (updates.py:30-36), Frobenius data term of the loss (base.py:197-198), with and without the stop criterion."""
    g = golden("f14_frobenius_fit")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        X, W0, H0 = g[f"{name}_X"], g[f"{name}_W0"], g[f"{name}_H0"]
        kw = dict(c["kw"])
        if not c.get("stop"):
            kw.update(tol=0, no_stop_criterion=True)
        r = oc.fit(X, c["k"], G=g.get(f"{name}_G"), W=W0.copy(), H=H0.copy(), shape_2d=shape, algo="l2_surrogate", l2=True,
                   max_iter=c["iters"], **kw)
        assert r["n_iter"] == int(g[f"{name}_n_iter"]), name
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-7, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-6, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=2e-5, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=2e-5, atol=1e-8, err_msg=name)


def test_f15_gradients_and_q_step(golden):
    """gradW / gradH (KL and l2 branches, vector mu, Laplacian) and update_q (updates.py:225-230, :303-342)."""
    g = golden("f15_gradients")
    for name in g["names"]:
        X, G, W0, H0, mu = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "mu"))
        L = oc.laplacian_matrix(*(int(v) for v in g[f"{name}_shape"]))
        np.testing.assert_allclose(oc.gradW(X, G, W0, H0), g[f"{name}_gradW"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(oc.gradW(X, G, W0, H0, l2=True), g[f"{name}_gradW_l2"], rtol=1e-11, atol=1e-10)
        # (the reference's L is a float32 matrix and `lambda_L * L` rounds 0.8 * 4 to float32: 6e-8 relative)
        np.testing.assert_allclose(oc.gradH(X, G, W0, H0, mu=mu, lambda_L=0.8, L=L, epsilon_reg=0.7), g[f"{name}_gradH"],
                                   rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(oc.gradH(X, G, W0, H0), g[f"{name}_gradH_plain"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(oc.gradH(X, G, W0, H0, mu=0.2, lambda_L=0.5, L=L, l2=True), g[f"{name}_gradH_l2"],
                                   rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(oc.update_q(G @ W0, H0), g[f"{name}_Q"], rtol=1e-12, atol=1e-15)


def test_f18_projected_gradient_steps_with_the_frobenius_gradient(golden):
    """proj_grad_step_w / _h called with l2=True (updates.py:353-395; the branch only a direct call reaches): dictionary and identity G,
    with / without the simplex over H, regularisers, fixed entries.  (The multiplier's bisection stops at dicotomy_tol = 1e-5: the
    simplex cases agree to that.)"""
    g = golden("f18_projected_gradient_l2")
    for name in g["names"]:
        X, G, W0, H0, mu = (g[f"{name}_{v}"] for v in ("X", "G", "W0", "H0", "mu"))
        gh, gw = (float(v) for v in g[f"{name}_gamma"])
        fW, fH = g[f"{name}_fixed_W"], g[f"{name}_fixed_H"]
        L = oc.laplacian_matrix(*(int(v) for v in g[f"{name}_shape"]))
        np.testing.assert_allclose(oc.proj_grad_step_w(X, G, W0, H0, gw, simplex_W=False, l2=True), g[f"{name}_W_l2"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(oc.proj_grad_step_w(X, G, W0, H0, gw, simplex_W=False, l2=True, fixed_W=fW), g[f"{name}_W_l2_fixed"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(oc.proj_grad_step_h(X, G, W0, H0, gh, simplex_H=True, l2=True), g[f"{name}_H_l2"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(oc.proj_grad_step_h(X, G, W0, H0, gh, simplex_H=False, mu=mu, lambda_L=0.6, L=L, epsilon_reg=0.8, l2=True, fixed_H=fH),
                                   g[f"{name}_H_l2_free"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(oc.proj_grad_step_h(X, G, W0, H0, gh, simplex_H=True, mu=0.2, lambda_L=0.5, L=L, l2=True), g[f"{name}_H_l2_reg"], rtol=0, atol=3e-5)


def test_f12_projected_gradient(golden):
    """algo="projected_gradient" with a given gamma = [gamma_H, gamma_W] (updates.py:317-395, dicotomy.py:84-108)."""
    g = golden("f12_projected_gradient")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        X, W0, H0 = g[f"{name}_X"], g[f"{name}_W0"], g[f"{name}_H0"]
        G = g.get(f"{name}_G")
        Gd = np.eye(c["n"]) if G is None else G
        kw = dict(c["kw"])
        gh, gw = kw["gamma"]
        Hs = oc.proj_grad_step_h(X, Gd, W0, H0.copy(), gh, simplex_H=kw["simplex_H"], mu=kw["mu"], lambda_L=kw["lambda_L"],
                                 L=oc.laplacian_matrix(*shape))
        np.testing.assert_allclose(Hs, g[f"{name}_step_H"], rtol=1e-9, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(oc.proj_grad_step_w(X, Gd, W0.copy(), H0, gw, simplex_W=False), g[f"{name}_step_W"], rtol=1e-10,
                                   atol=1e-16, err_msg=name)
        r = oc.fit(X, c["k"], G=G, W=W0.copy(), H=H0.copy(), shape_2d=shape, algo="projected_gradient", tol=0,
                   no_stop_criterion=True, max_iter=c["iters"], **kw)
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-7, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-6, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=2e-5, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=2e-5, atol=1e-8, err_msg=name)


def test_f13_projected_gradient_linesearch(golden):
    """The projected gradient's own linesearch (smooth_nmf.py:382-401, :438-447): gamma_H and gamma_W follow the quadratic bound."""
    g = golden("f13_projected_gradient_linesearch")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        r = oc.fit(g[f"{name}_X"], c["k"], G=g.get(f"{name}_G"), W=g[f"{name}_W0"].copy(), H=g[f"{name}_H0"].copy(), shape_2d=shape,
                   algo="projected_gradient", linesearch=True, tol=0, no_stop_criterion=True, max_iter=c["iters"], **c["kw"])
        np.testing.assert_allclose(r["gammas"], g[f"{name}_gammas"], rtol=1e-12, err_msg=name)
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-6, err_msg=name)
        if name == "lx":
            assert (np.diff(g[f"{name}_gammas"][:, 0]) > 0).any(), "the fixture must contain an increase of gamma"


def test_sparse_oracle_on_f6(golden):
    """The sparse-aware fp64 variant (oracle/mu_oracle_sparse.py: Y only at the non-zero entries of X, sum(Y) by
    colsum . rowsum, G^T (R H^T)) that the full-size GPU parity tests use is pinned here: on every F6 problem it must
    reproduce the reference-generated trajectories (free mode) like the faithful oracle does."""
    from oracle import mu_oracle_sparse as osp
    g = golden("f6_trajectories")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        G = g.get(f"{name}_G")
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        kw = dict(c["kw"])
        if kw.pop("normalize", False):
            continue
        X_ = oc.remove_zeros_lines(np.asarray(g[f"{name}_X"]), oc.LOG_SHIFT)
        _, W0, H0 = oc.initialize_algorithms(X_, G, g[f"{name}_W0"].copy(), g[f"{name}_H0"].copy(), c["k"], None, None,
                                             kw.get("simplex_H", False), kw.get("simplex_W", True))
        r = osp.fit(osp.SparseX.from_dense(X_), c["k"], G=G, W=W0, H=H0, shape_2d=shape, record_at=(1, 2, 5, 50),
                    tol=0, max_iter=50, **kw)
        pre = f"{name}_free"
        np.testing.assert_allclose(r["losses"], g[f"{pre}_losses"], rtol=1e-9, err_msg=pre)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{pre}_detailed"][:, :3], rtol=1e-9, atol=1e-18, err_msg=pre)
        np.testing.assert_allclose(r["rel"], g[f"{pre}_rel"], rtol=1e-7, atol=1e-12, err_msg=pre)
        Wf, Hf = r["W"], r["H"]
        if not kw.get("simplex_H", False) and not kw.get("simplex_W", True):
            Wf, Hf = oc.rescaled_DH(Wf, Hf)      # base.py:399-400, outside the loop
        np.testing.assert_allclose(Wf, g[f"{pre}_W"], rtol=1e-8, atol=1e-14, err_msg=pre)
        np.testing.assert_allclose(Hf, g[f"{pre}_H"], rtol=1e-8, atol=1e-14, err_msg=pre)
        for t, (Wt, Ht) in r["snapshots"].items():
            np.testing.assert_allclose(Wt, g[f"{pre}_W{t}"], rtol=1e-8, atol=1e-14)
            np.testing.assert_allclose(Ht, g[f"{pre}_H{t}"], rtol=1e-8, atol=1e-14)
        assert osp.dropped_eps_logy(osp.SparseX.from_dense(X_), G, r["W"], r["H"]) < 1e-9 * abs(r["losses"][-1]) * X_.size


def test_f16_physics_model_refreshes_g(golden):
    """Fixture F16: fits with a PhysicalModel whose G depends on W (tests/physics_double.py mixed into the reference's
    abstract class by the generator): NMF_update every third iteration, the loss re-evaluated with the new G before the
    next stop test (base.py:388-392), simplex over the NMF_simplex() rows only, and the Frobenius W step with a live G."""
    from physics_double import AbsorbingModel
    g = golden("f16_physics_model")
    cfgs = json.loads(str(g["configs"]))
    for name in g["names"]:
        c = cfgs[name]
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        model = AbsorbingModel(g[f"{name}_G0"], g[f"{name}_Abs"], float(g[f"{name}_strength"]), c["m0"])
        kw = dict(c["kw"])
        if not c.get("stop"):
            kw.update(tol=0, no_stop_criterion=True)
        r = oc.fit(g[f"{name}_X"], c["k"], W=g[f"{name}_W0"].copy(), H=g[f"{name}_H0"].copy(), shape_2d=shape, max_iter=c["iters"],
                   physics_model=model, **kw)
        assert r["n_iter"] == int(g[f"{name}_n_iter"]) and model.updates == int(g[f"{name}_updates"]), name
        np.testing.assert_allclose(r["losses"], g[f"{name}_losses"], rtol=1e-9, err_msg=name)
        np.testing.assert_allclose(r["detailed_losses"], g[f"{name}_detailed"], rtol=1e-9, atol=1e-18, err_msg=name)
        np.testing.assert_allclose(r["rel"], g[f"{name}_rel"], rtol=1e-7, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(r["W"], g[f"{name}_W"], rtol=1e-8, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(r["H"], g[f"{name}_H"], rtol=1e-8, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(r["G"], g[f"{name}_G"], rtol=1e-10, err_msg=name)


def test_f17_hyperspy_calling_convention(golden):
    """Fixture F17: the reference run the way hyperspy's decomposition(algorithm=est) runs it - hspy_comp=True, the (pixels,
    channels) matrix of a 96 x 96 x 512 cube (base.py:243-247, :412-420; eds_spim.py:597-604) - without and with the stop
    rules.  The oracle on the transposed matrix gives the same factors: loadings = H^T, components_ = (G W)^T."""
    g = golden("f17_hyperspy_ingest")
    Xp = g["X_u8"].astype(np.float64)                 # (p, n)
    nx, ny = (int(v) for v in g["shape"])
    k = g["W0"].shape[1]
    for tag, kw in (("free", dict(tol=0, no_stop_criterion=True, max_iter=30)), ("stop", dict(tol=6e-4, max_iter=200))):
        r = oc.fit(np.ascontiguousarray(Xp.T), k, W=g["W0"].copy(), H=g["H0"].copy(), shape_2d=(nx, ny), simplex_H=True, simplex_W=False,
                   lambda_L=1.0, **kw)
        assert r["n_iter"] == int(g[f"{tag}_n_iter"]), tag
        np.testing.assert_allclose(r["losses"], g[f"{tag}_losses"], rtol=1e-10)
        np.testing.assert_allclose(r["H"].T, g[f"{tag}_loadings"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(r["GW"].T, g[f"{tag}_components"], rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(r["rel"], g[f"{tag}_rel"], rtol=1e-6)
    # the stop decision is not a coin toss for an fp32 path: the decrease that ends the run clears the tolerance by > 2 % of it,
    # and so do the decreases before it on the other side (base.py:362: |eval_before - eval_after| / eval_init < tol)
    ls = np.concatenate([[r["eval_init"]], g["stop_losses"]])
    dec = np.abs(np.diff(ls)) / abs(r["eval_init"])
    assert dec[-1] < 6e-4 * 0.98 and (dec[:-1] > 6e-4 * 1.02).all(), dec


def _fuzz_draws(golden):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_fuzz as fz
    g = golden("f21_reference_on_the_fuzz_draws")
    for tag, status in json.loads(str(g["index"])):
        wide, seed = int(tag[1]), int(tag.split("_s")[1])
        yield tag, status, (lambda w=wide, sd=seed: fz._case(sd, wide=w)), g


def test_oracle_fit_equals_the_reference_on_the_fuzz_draws(golden):
    """Fixture F21: the REFERENCE's SmoothNMF on 48 draws of the randomised sweep (all four solvers, linesearch, dictionaries, fixed entries,
    mu vectors, lines without counts, 1..32 components).  The oracle's fit loop in its reference-faithful mode (the reference's own bisection,
    not the converged root) reproduces the losses to 1e-9 and the factors to fp32 storage - and refuses / diverges on the draws the reference
    refuses / diverges on."""
    n_ok = 0
    for tag, status, make, g in _fuzz_draws(golden):
        c = make()
        if status == "refused":
            with pytest.raises(AssertionError):
                oc.fit(c["X"], c["k"], G=c["G"], W=c["W0"].copy(), H=c["H0"].copy(), shape_2d=c["shape"], algo=c["algo"], tol=0,
                       no_stop_criterion=True, max_iter=6, **c["kw"], **c["extra"])
            continue
        if status != "ok":
            continue
        assert float(c["X"].sum()) == float(g[f"{tag}_x_sum"]), tag       # the draw regenerated here is the one the reference saw
        ref = oc.fit(c["X"], c["k"], G=c["G"], W=c["W0"].copy(), H=c["H0"].copy(), shape_2d=c["shape"], algo=c["algo"], tol=0,
                     no_stop_criterion=True, max_iter=6, **c["kw"], **c["extra"])
        np.testing.assert_allclose(ref["losses"], g[f"{tag}_losses"], rtol=1e-9, err_msg=tag)
        np.testing.assert_allclose(ref["H"], g[f"{tag}_H"], rtol=2e-6, atol=1e-7, err_msg=tag)
        np.testing.assert_allclose(ref["W"], g[f"{tag}_W"], rtol=2e-6, atol=1e-7 * np.abs(g[f"{tag}_W"]).max(), err_msg=tag)
        n_ok += 1
    assert n_ok >= 40

