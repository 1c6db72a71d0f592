#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the UNMODIFIED reference.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference.  The reference imports ``exspy`` (absent here) at module import time
(espm/conf.py:2-4, espm/utils.py:8) without ever touching it on the multiplicative-update
path, so an empty stand-in package is put on sys.path from a temp dir (outside the repo).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Only DATA is stored: inputs and the reference's outputs.  Each archive also records the
library versions used.  Fixture families follow SURVEY.md section 8(c): F1..F8; F17: the hyperspy calling convention on a
(96 x 96, 512) cube; F9 covers the options of section 8(f) rank 4 that
are built (linesearch, true_D / true_H tracking).
"""
import contextlib
import io
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ESPM_REFERENCE", "/root/reference")


def _install_exspy_stub():
    root = tempfile.mkdtemp(prefix="exspy_stub_")
    for sub in ("exspy", "exspy/_misc", "exspy/_misc/eds"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        open(os.path.join(root, sub, "__init__.py"), "w").close()
    with open(os.path.join(root, "exspy/_misc/eds/ffast_mac.py"), "w") as f:
        f.write("ffast_mac = {}\n")
    with open(os.path.join(root, "exspy/material.py"), "w") as f:
        f.write("def atomic_to_weight(*a, **k):\n    raise NotImplementedError\n"
                "def density_of_mixture(*a, **k):\n    raise NotImplementedError\n")
    sys.path.insert(0, root)


_install_exspy_stub()
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import sklearn  # noqa: E402
from espm.conf import dicotomy_tol, log_shift  # noqa: E402
from espm.estimators import SmoothNMF  # noqa: E402
from espm.estimators.base import normalization_factor  # noqa: E402
from espm.estimators.dicotomy import dichotomy_simplex  # noqa: E402
from espm.estimators.updates import (initialize_algorithms, multiplicative_step_h,  # noqa: E402
                                     multiplicative_step_w)
from espm.measures import Frobenius_loss, KLdiv_loss, log_reg, trace_xtLx  # noqa: E402
from espm.utils import create_laplacian_matrix, rescaled_DH  # noqa: E402

VERSIONS = json.dumps({"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__,
                       "reference": "adriente/espm v1.1.3 (2025-02-05)"})


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, versions=np.array(VERSIONS), **arrays)
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def synth(rng, n, nx, ny, k, m=None, counts=40.0, poisson=True):
    """Small Poisson spectrum image: X ~ Poisson(G W H) with H on the simplex."""
    p = nx * ny
    H = rng.random((k, p)) ** 2 + 0.05
    H /= H.sum(axis=0, keepdims=True)
    if m is None:
        G = None
        W = rng.random((n, k)) ** 3 * counts / n * 4 + 1e-3
        D = W
    else:
        G = rng.random((n, m)) * (rng.random((n, m)) < 0.4) + 0.01
        W = rng.random((m, k)) * counts / n
        D = G @ W
    Y = D @ H
    X = rng.poisson(Y).astype(np.float64) if poisson else Y
    return X, G, W, H


# ------------------------------------------------------------------ F1: dichotomy_simplex
def f1():
    out = {}
    # known-answer vectors of espm/tests/test_updates.py:131-141 and :146-151
    kat = [
        (np.array([[1, 1, 0, 0, 0, 2]], float).T, np.array([[1, 1, 3, 5, 4, 2]], float).T, 0.0, 1e-6),
        (np.array([[1, 1, 0, 0, 0, 2]], float).T, np.array([[1, 1, 0, 5, 0, 2]], float).T, 0.0, 1e-6),
        (np.array([[3, 0.5]], float).T, np.array([[1, 1]], float).T, 0.25, 1e-8),
    ]
    for i, (num, den, eps, tol) in enumerate(kat):
        out[f"kat{i}_num"], out[f"kat{i}_den"] = num, den
        out[f"kat{i}_eps"], out[f"kat{i}_tol"] = np.array(eps), np.array(tol)
        out[f"kat{i}_nu"] = dichotomy_simplex(num.copy(), den.copy(), eps, tol=tol)
    rng = np.random.default_rng(101)
    span = np.logspace(-6, 6, 17)
    k, p = 5, 64
    case = 0
    for eps in (0.0, 0.02):
        for den_cols in (p, 1):
            for zeros in (False, True):
                num = rng.choice(span, (k, p)) * rng.random((k, p))
                if zeros:
                    num[np.tile(np.arange(k), p // k + 1)[:p], np.arange(p)] = 0
                den = rng.choice(span, (k, den_cols)) * rng.random((k, den_cols))
                for tol, tag in ((dicotomy_tol, "t5"), (0.0, "t0")):
                    nu = dichotomy_simplex(num.copy(), den.copy(), eps, tol=tol, maxit=100)
                    out[f"rnd{case}_{tag}_nu"] = nu
                out[f"rnd{case}_num"], out[f"rnd{case}_den"] = num, den
                out[f"rnd{case}_eps"] = np.array(eps)
                case += 1
    out["n_rnd"] = np.array(case)
    out["n_kat"] = np.array(len(kat))
    save("f1_dichotomy", **out)


# ------------------------------------------------------------------ F2: multiplicative_step_h
def f2():
    rng = np.random.default_rng(202)
    n, nx, ny, k, m = 40, 6, 9, 4, 7
    out, case = {}, 0
    L = create_laplacian_matrix(nx, ny)
    mu_vec = np.array([0.0, 0.7, 1.3, 0.2])
    for with_G in (False, True):
        X, G, W, H = synth(rng, n, nx, ny, k, m if with_G else None)
        Gd = np.diag(np.ones(n)) if G is None else G
        H0 = rng.random((k, nx * ny)) + 0.01
        H0 /= H0.sum(axis=0, keepdims=True)
        W0 = W * (0.5 + rng.random(W.shape))
        fixed = -np.ones_like(H0)
        fixed[0, :5] = 0.3
        fixed[1, :5] = 0.0
        out[f"in{int(with_G)}_X"], out[f"in{int(with_G)}_G"] = X, Gd
        out[f"in{int(with_G)}_W"], out[f"in{int(with_G)}_H"] = W0, H0
        out[f"in{int(with_G)}_fixed"] = fixed
        for simplex in (False, True):
            for lam in (0.0, 2.0):
                for mu_on in (False, True):
                    for fix_on in (False, True):
                        mu = mu_vec if mu_on else 0
                        Hn = multiplicative_step_h(X, Gd, W0, H0.copy(), simplex_H=simplex, mu=mu,
                                                   log_shift=log_shift, epsilon_reg=0.8, safe=True,
                                                   dicotomy_tol=dicotomy_tol, lambda_L=lam, L=L,
                                                   fixed_H=fixed if fix_on else None)
                        out[f"c{case}_H"] = Hn
                        out[f"c{case}_cfg"] = np.array([int(with_G), int(simplex), lam, int(mu_on), int(fix_on)])
                        case += 1
        # l2 branch, reachable only by direct call (updates.py:109-118)
        out[f"l2_{int(with_G)}_H"] = multiplicative_step_h(X, Gd, W0, H0.copy(), simplex_H=True, l2=True)
        # scalar mu and a non-default sigmaL
        out[f"smu_{int(with_G)}_H"] = multiplicative_step_h(X, Gd, W0, H0.copy(), simplex_H=True, mu=0.4,
                                                            lambda_L=0.5, L=L, sigmaL=11.0)
        # identity "Laplacian" used when shape_2d is None (base.py:289-291)
        from scipy.sparse import lil_matrix
        Lid = lil_matrix((nx * ny, nx * ny), dtype=np.float32)
        Lid.setdiag([1] * (nx * ny))
        out[f"lid_{int(with_G)}_H"] = multiplicative_step_h(X, Gd, W0, H0.copy(), simplex_H=True,
                                                            lambda_L=1.5, L=Lid)
    out["n_cases"] = np.array(case)
    out["mu_vec"], out["shape_2d"], out["epsilon_reg"] = mu_vec, np.array([nx, ny]), np.array(0.8)
    save("f2_step_h", **out)


# ------------------------------------------------------------------ F3: multiplicative_step_w
def f3():
    rng = np.random.default_rng(303)
    n, nx, ny, k, m = 40, 6, 9, 4, 7
    out, case = {}, 0
    for with_G in (False, True):
        X, G, W, H = synth(rng, n, nx, ny, k, m if with_G else None)
        Gd = np.diag(np.ones(n)) if G is None else G
        W0 = W * (0.5 + rng.random(W.shape))
        fixed = -np.ones_like(W0)
        fixed[0, 0] = 0.0
        fixed[2, 1] = 0.25
        t = int(with_G)
        out[f"in{t}_X"], out[f"in{t}_G"], out[f"in{t}_W"], out[f"in{t}_H"] = X, Gd, W0, H
        out[f"in{t}_fixed"] = fixed
        for simplex in (False, True):
            for fix_on in (False, True):
                Wn = multiplicative_step_w(X, Gd, W0.copy(), H, simplex_W=simplex, log_shift=log_shift,
                                           safe=True, fixed_W=fixed if fix_on else None)
                out[f"c{case}_W"] = Wn
                out[f"c{case}_cfg"] = np.array([t, int(simplex), int(fix_on)])
                case += 1
        out[f"l2_{t}_W"] = multiplicative_step_w(X, Gd, W0.copy(), H, simplex_W=False, l2=True)
    out["n_cases"] = np.array(case)
    save("f3_step_w", **out)


# ------------------------------------------------------------------ F4: Laplacian
def f4():
    rng = np.random.default_rng(404)
    out = {}
    shapes = [(2, 2), (3, 7), (8, 4), (16, 16)]
    for i, (nx, ny) in enumerate(shapes):
        L = create_laplacian_matrix(nx, ny)
        H = rng.random((3, nx * ny))
        out[f"s{i}_H"] = H
        out[f"s{i}_HL"] = H @ L
        out[f"s{i}_trace"] = np.array(trace_xtLx(L, H.T))
        if nx * ny <= 64:
            out[f"s{i}_dense"] = np.asarray(L.todense())
    out["shapes"] = np.array(shapes)
    save("f4_laplacian", **out)


# ------------------------------------------------------------------ F5: losses
def f5():
    rng = np.random.default_rng(505)
    n, nx, ny, k = 30, 5, 8, 3
    X, G, W, H = synth(rng, n, nx, ny, k)
    X[3, :] = 0  # an all-zero channel: exercised by remove_zeros_lines
    mu = np.array([0.0, 0.5, 2.0])
    out = dict(X=X, W=W, H=H, mu=mu)
    out["KLdiv_loss"] = np.array(KLdiv_loss(X, W, H, log_shift))
    out["KLdiv_loss_avg"] = np.array(KLdiv_loss(X, W, H, log_shift, average=True))
    out["Frobenius_loss"] = np.array(Frobenius_loss(X, W, H))
    out["log_reg"] = np.array(log_reg(H, mu, 0.8))
    out["log_reg_scalar"] = np.array(log_reg(H, 0.3, 1))
    for avg in (True, False):
        est = SmoothNMF(n_components=k, lambda_L=1.5, mu=mu, epsilon_reg=0.8, shape_2d=(nx, ny),
                        simplex_H=True, simplex_W=False, max_iter=1, verbose=0)
        quiet(est.fit_transform, X, W=W.copy(), H=H.copy())
        # loss of an arbitrary state through the fitted estimator (uses est.X_, est.G_, est.L_)
        val = est.loss(W, H, average=avg)
        out[f"loss_avg{int(avg)}"] = np.array(val)
        out[f"detailed_avg{int(avg)}"] = np.array(est.detailed_loss_, dtype=float)
        out["X_"] = est.X_
        out["const_KL"] = np.array(est.const_KL_)
    out["shape_2d"] = np.array([nx, ny])
    save("f5_losses", **out)


# ------------------------------------------------------------------ F6: trajectories
TRAJ = {
    # scaled-down analogues of BASELINE.json configs 1, 2, 3, 5
    "c1": dict(n=24, nx=6, ny=5, k=3, m=None, kw=dict(simplex_H=False, simplex_W=False, lambda_L=0.0, mu=0)),
    "c2": dict(n=48, nx=8, ny=8, k=3, m=None, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.0, mu=0)),
    "c3": dict(n=64, nx=12, ny=10, k=5, m=None, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0)),
    "c5": dict(n=60, nx=10, ny=12, k=4, m=9, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0.05)),
    "cw": dict(n=32, nx=6, ny=6, k=3, m=None, kw=dict(simplex_H=False, simplex_W=True, lambda_L=0.5, mu=0)),
}


def f6():
    rng = np.random.default_rng(606)
    out = {}
    for name, c in TRAJ.items():
        X, G, W, H = synth(rng, c["n"], c["nx"], c["ny"], c["k"], c["m"])
        p = c["nx"] * c["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c["nx"], c["ny"]])
        for mode, extra in (("free", dict(tol=0, no_stop_criterion=True, max_iter=50)),
                            ("stop", dict(tol=1e-3, max_iter=200))):
            snaps = {}

            class Rec(SmoothNMF):
                def _iteration(self, W, H):
                    W, H = super()._iteration(W, H)
                    if self.n_iter_ + 1 in (1, 2, 5, 50):
                        snaps[self.n_iter_ + 1] = (W.copy(), H.copy())
                    return W, H

            est = Rec(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, **c["kw"], **extra)
            GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
            pre = f"{name}_{mode}"
            out[f"{pre}_GW"], out[f"{pre}_W"], out[f"{pre}_H"] = GW, est.W_, est.H_
            out[f"{pre}_losses"] = np.array(est.losses_)
            out[f"{pre}_detailed"] = np.array(est.detailed_losses_, dtype=float)
            out[f"{pre}_rel"] = np.array(est.rel_)
            out[f"{pre}_n_iter"] = np.array(est.n_iter_)
            out[f"{pre}_recon"] = np.array(est.reconstruction_err_)
            if mode == "free":
                for t, (Wt, Ht) in snaps.items():
                    out[f"{pre}_W{t}"], out[f"{pre}_H{t}"] = Wt, Ht
    out["names"] = np.array(list(TRAJ))
    out["configs"] = np.array(json.dumps({k: {**v, "kw": v["kw"]} for k, v in TRAJ.items()}))
    save("f6_trajectories", **out)


# ------------------------------------------------------------------ F7: initialize_algorithms
def f7():
    rng = np.random.default_rng(707)
    X, G, W, H = synth(rng, 36, 6, 7, 3, 8)
    out = dict(X=X, G=G)
    for init in (None, "random", "nndsvd"):
        for use_G in (False, True):
            for simplex_H in (False, True):
                G_, W_, H_ = initialize_algorithms(X, G if use_G else None, None, None, n_components=3,
                                                   init=init, random_state=0, simplex_H=simplex_H,
                                                   simplex_W=not simplex_H)
                tag = f"{init}_{int(use_G)}_{int(simplex_H)}"
                out[f"{tag}_W"], out[f"{tag}_H"] = W_, H_
    # warm starts: only W or only H given (updates.py:188-189, :213-218)
    G_, W_, H_ = initialize_algorithms(X, G, W, None, 3, None, 0, True, False)
    out["Wgiven_W0"], out["Wgiven_H"] = W, H_
    G_, W_, H_ = initialize_algorithms(X, G, None, H, 3, None, 0, True, False)
    out["Hgiven_H0"], out["Hgiven_W"] = H, W_
    D, Hr = rescaled_DH(G @ W * 3.0, H / 3.0)
    out["rescaled_D"], out["rescaled_H"] = D, Hr
    save("f7_init", **out)


# ------------------------------------------------------------------ F8: API behaviour
def f8():
    out = {}
    est = quiet(SmoothNMF)
    params = est.get_params()
    out["default_params"] = np.array(json.dumps({k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v))
                                                 for k, v in params.items()}, sort_keys=True))
    e2 = quiet(SmoothNMF, simplex_H=True, simplex_W=True, l2=True, lambda_L=-1, algo="nope", epsilon_reg=0)
    out["coerced"] = np.array(json.dumps(dict(simplex_H=e2.simplex_H, simplex_W=e2.simplex_W, l2=e2.l2,
                                              lambda_L=e2.lambda_L, algo=e2.algo, epsilon_reg=e2.epsilon_reg)))
    e3 = quiet(SmoothNMF, linesearch=True, lambda_L=0.0)
    out["coerced_linesearch"] = np.array(json.dumps(dict(lambda_L=e3.lambda_L, linesearch=e3.linesearch)))
    rng = np.random.default_rng(808)
    X, G, W, H = synth(rng, 20, 4, 5, 2)
    est = SmoothNMF(n_components=2, max_iter=3, simplex_H=True, simplex_W=False, verbose=0, hspy_comp=True)
    ret = quiet(est.fit_transform, X.T.copy(), W=W.copy(), H=H.copy())
    out["hspy_X"], out["hspy_W0"], out["hspy_H0"] = X, W, H
    out["hspy_ret"], out["hspy_components"] = ret, est.components_
    out["loss_names"] = np.array(json.dumps(list(est.get_losses().dtype.names)))
    out["get_losses"] = np.array(est.get_losses().tolist())
    out["inverse_transform"] = est.inverse_transform(est.W_)
    # normalize=True scale invariance inputs/outputs (espm/tests/test_estimators.py:207-247)
    rs = np.random.RandomState(0)
    Xn = rs.rand(10, 32) * 100
    fac = rs.rand() * 50 + 0.1
    est = SmoothNMF(n_components=5, lambda_L=1.0, max_iter=10, init="nndsvd", normalize=True, shape_2d=[8, 4],
                    random_state=0, simplex_W=False, simplex_H=True, verbose=0)
    GP = quiet(est.fit_transform, Xn)
    out["norm_X"], out["norm_fac"], out["norm_GP"], out["norm_H"] = Xn, np.array(fac), GP, est.H_
    out["norm_W"], out["norm_factor"] = est.W_, np.array(normalization_factor(est.remove_zeros_lines(Xn, log_shift), 5))
    # no simplex at all -> rescaled_DH post-processing (base.py:399-400)
    est = SmoothNMF(n_components=2, max_iter=4, simplex_H=False, simplex_W=False, verbose=0)
    GW = quiet(est.fit_transform, X, W=W.copy(), H=H.copy())
    out["nosimplex_GW"], out["nosimplex_W"], out["nosimplex_H"] = GW, est.W_, est.H_
    out["nosimplex_recon"] = np.array(est.reconstruction_err_)
    save("f8_api", **out)


# ------------------------------------------------------------------ F9: linesearch and truth tracking
LS = {
    # lambda_L != 1 on purpose: diff_surrogate is called with its default lambda_L = 1 (smooth_nmf.py:377)
    "ls3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=40, kw=dict(simplex_H=True, simplex_W=False, lambda_L=2.0, mu=0)),
    "ls5": dict(n=60, nx=10, ny=12, k=4, m=9, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0.05)),
    "lsw": dict(n=32, nx=6, ny=6, k=3, m=None, iters=30, kw=dict(simplex_H=False, simplex_W=True, lambda_L=0.5, mu=0, gamma=3.0)),
}
TM = {
    "tm_h": dict(n=48, nx=8, ny=8, k=3, m=None, iters=12, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.5, mu=0)),
    "tm_0": dict(n=40, nx=6, ny=7, k=3, m=7, iters=10, kw=dict(simplex_H=False, simplex_W=False, lambda_L=0.0, mu=0.02)),
}


def f9():
    rng = np.random.default_rng(909)
    out = {}
    for name, c in {**LS, **TM}.items():
        X, G, W, H = synth(rng, c["n"], c["nx"], c["ny"], c["k"], c["m"])
        p = c["nx"] * c["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c["nx"], c["ny"]])
        extra = dict(tol=0, no_stop_criterion=True, max_iter=c["iters"])
        if name in LS:
            est = SmoothNMF(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, linesearch=True, **c["kw"], **extra)
        else:
            true_D = W if G is None else G @ W
            out[f"{name}_true_D"], out[f"{name}_true_H"] = true_D, H
            est = SmoothNMF(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, true_D=true_D, true_H=H,
                            **c["kw"], **extra)
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
        if name in TM:
            out[f"{name}_angles"] = np.array(est.angles_, dtype=float)
            out[f"{name}_mse"] = np.array(est.mse_, dtype=float)
            out[f"{name}_true_losses"] = np.array(est.true_losses_, dtype=float)
            gl = est.get_losses()
            out[f"{name}_loss_names"] = np.array(json.dumps(list(gl.dtype.names)))
            out[f"{name}_get_losses"] = np.array(gl.tolist())
    out["names_ls"] = np.array(list(LS))
    out["names_tm"] = np.array(list(TM))
    out["configs"] = np.array(json.dumps({**LS, **TM}))
    save("f9_linesearch_truth", **out)


# ------------------------------------------------------------------ F10: the Bregman variant (algo="bmd", use_bregman=True)
BMD = {
    "b3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0)),
    # (a non-square G makes the reference's own W step fail: np.allclose(G, np.eye(n)) at updates.py:43 cannot broadcast)
    "bm": dict(n=60, nx=10, ny=12, k=4, m=None, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0.05)),
    "b0": dict(n=32, nx=6, ny=6, k=3, m=None, iters=20, kw=dict(simplex_H=False, simplex_W=False, lambda_L=0.0, mu=0)),
    "bl": dict(n=48, nx=8, ny=8, k=3, m=None, iters=25, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.7, mu=0, linesearch=True)),
}


def f10():
    rng = np.random.default_rng(1010)
    out = {}
    for name, c in BMD.items():
        X, G, W, H = synth(rng, c["n"], c["nx"], c["ny"], c["k"], c["m"])
        p = c["nx"] * c["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        Gd = np.eye(c["n"]) if G is None else G
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c["nx"], c["ny"]])
        L = create_laplacian_matrix(c["nx"], c["ny"])
        # single steps (direct calls)
        mu = c["kw"]["mu"]
        out[f"{name}_step_H"] = multiplicative_step_h(X, Gd, W0, H0.copy(), simplex_H=c["kw"]["simplex_H"], mu=mu,
                                                      lambda_L=c["kw"]["lambda_L"], L=L, sigmaL=8, use_bregman=True)
        out[f"{name}_step_W"] = multiplicative_step_w(X, Gd, W0.copy(), H0, simplex_W=False, use_bregman=True)
        est = SmoothNMF(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, algo="bmd", tol=0,
                        no_stop_criterion=True, max_iter=c["iters"], **c["kw"])
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
    out["names"] = np.array(list(BMD))
    out["configs"] = np.array(json.dumps(BMD))
    save("f10_bregman", **out)


# ------------------------------------------------------------------ F11: the quadratic ("l2") surrogate of the Laplacian term
HQ = {
    "q3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0)),
    "q5": dict(n=60, nx=10, ny=12, k=4, m=9, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.6)),
    "qw": dict(n=32, nx=6, ny=6, k=3, m=None, iters=20, kw=dict(simplex_H=False, simplex_W=True, lambda_L=0.5)),
    "q0": dict(n=40, nx=6, ny=7, k=3, m=None, iters=15, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.0)),
    "ql": dict(n=48, nx=8, ny=8, k=3, m=None, iters=25, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.7, linesearch=True)),
}


def f11():
    from espm.estimators.updates import multiplicative_step_hq
    from espm.estimators.dicotomy import dichotomy_simplex_acc
    rng = np.random.default_rng(1111)
    out = {}
    # root finder known answers: sum_k (sqrt((b + nu)^2 + 4 a c) - nu - b) / (2 a) = 1
    a = 3.7
    b = rng.standard_normal((5, 40)) * 2.0 + 1.0
    c = rng.random((5, 40)) * 3.0
    out["acc_a"], out["acc_b"], out["acc_c"] = np.array(a), b, c
    out["acc_nu"] = dichotomy_simplex_acc(a, b.copy(), c.copy(), log_shift=0.0, tol=1e-12, maxit=200)
    for name, c_ in HQ.items():
        X, G, W, H = synth(rng, c_["n"], c_["nx"], c_["ny"], c_["k"], c_["m"])
        p = c_["nx"] * c_["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c_["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        Gd = np.eye(c_["n"]) if G is None else G
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c_["nx"], c_["ny"]])
        L = create_laplacian_matrix(c_["nx"], c_["ny"])
        out[f"{name}_step_H"] = multiplicative_step_hq(X, Gd, W0, H0.copy(), simplex_H=c_["kw"]["simplex_H"],
                                                       lambda_L=c_["kw"]["lambda_L"], L=L, sigmaL=8)
        est = SmoothNMF(n_components=c_["k"], G=G, shape_2d=(c_["nx"], c_["ny"]), verbose=0, algo="l2_surrogate", tol=0,
                        no_stop_criterion=True, max_iter=c_["iters"], **c_["kw"])
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
    out["names"] = np.array(list(HQ))
    out["configs"] = np.array(json.dumps(HQ))
    save("f11_quadratic_surrogate", **out)


# ------------------------------------------------------------------ F12: projected gradient (no linesearch)
PG = {
    # gamma = [gamma_H, gamma_W]: the default (Lipschitz bounds at log_shift) is ~1e28 and freezes the iterates
    "p3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0, gamma=[400.0, 4000.0])),
    "p5": dict(n=60, nx=10, ny=12, k=4, m=9, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.6, mu=0.05, gamma=[2000.0, 20000.0])),
    "p0": dict(n=32, nx=6, ny=6, k=3, m=None, iters=20, kw=dict(simplex_H=False, simplex_W=False, lambda_L=0.0, mu=0, gamma=[300.0, 1500.0])),
}


def f12():
    from espm.estimators.updates import proj_grad_step_h, proj_grad_step_w
    rng = np.random.default_rng(1212)
    out = {}
    for name, c in PG.items():
        X, G, W, H = synth(rng, c["n"], c["nx"], c["ny"], c["k"], c["m"])
        p = c["nx"] * c["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        Gd = np.eye(c["n"]) if G is None else G
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c["nx"], c["ny"]])
        L = create_laplacian_matrix(c["nx"], c["ny"])
        gh, gw = c["kw"]["gamma"]
        out[f"{name}_step_H"] = proj_grad_step_h(X, Gd, W0, H0.copy(), gh, simplex_H=c["kw"]["simplex_H"], mu=c["kw"]["mu"],
                                                 lambda_L=c["kw"]["lambda_L"], L=L)
        out[f"{name}_step_W"] = proj_grad_step_w(X, Gd, W0.copy(), H0, gw, simplex_W=False)
        est = SmoothNMF(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, algo="projected_gradient", tol=0,
                        no_stop_criterion=True, max_iter=c["iters"], **c["kw"])
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
    out["names"] = np.array(list(PG))
    out["configs"] = np.array(json.dumps(PG))
    save("f12_projected_gradient", **out)


# ------------------------------------------------------------------ F13: projected gradient WITH its linesearch
PGL = {
    "l3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=40, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0, gamma=[400.0, 4000.0])),
    "l5": dict(n=60, nx=10, ny=12, k=4, m=9, iters=40, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.6, mu=0.05, gamma=[2000.0, 20000.0])),
    "l0": dict(n=32, nx=6, ny=6, k=3, m=None, iters=30, kw=dict(simplex_H=False, simplex_W=False, lambda_L=0.5, mu=0, gamma=[300.0, 1500.0])),
    # small gammas: at iteration 11 an entry of GWH reaches the clamp, the loss jumps to ~5e9 and gamma goes UP (x 1.5)
    "lx": dict(n=40, nx=7, ny=6, k=3, m=None, iters=12, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.4, mu=0, gamma=[25.0, 120.0])),
}


def f13():
    rng = np.random.default_rng(1313)
    out = {}
    for name, c in PGL.items():
        X, G, W, H = synth(rng, c["n"], c["nx"], c["ny"], c["k"], c["m"])
        p = c["nx"] * c["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c["nx"], c["ny"]])
        gam = []

        class Rec(SmoothNMF):
            def _iteration(self, W, H):
                W, H = super()._iteration(W, H)
                gam.append(list(self.gamma_))
                return W, H

        est = Rec(n_components=c["k"], G=G, shape_2d=(c["nx"], c["ny"]), verbose=0, algo="projected_gradient", linesearch=True, tol=0,
                  no_stop_criterion=True, max_iter=c["iters"], **c["kw"])
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_gammas"] = np.array(gam, dtype=float)
    out["names"] = np.array(list(PGL))
    out["configs"] = np.array(json.dumps(PGL))
    save("f13_projected_gradient_linesearch", **out)


# ------------------------------------------------------------------ F14: Frobenius data term inside a fit (l2=True)
L2FIT = {
    # l2 survives only with algo="l2_surrogate" and no linesearch (smooth_nmf.py:223-237): quadratic-surrogate H step,
    # Frobenius W step (no simplex over W there), loss = half the squared Frobenius distance + the regularisations
    "l3": dict(n=64, nx=12, ny=10, k=5, m=None, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0)),
    "l5": dict(n=60, nx=10, ny=12, k=4, m=9, iters=30, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.6)),
    "lw": dict(n=32, nx=6, ny=6, k=3, m=None, iters=20, kw=dict(simplex_H=False, simplex_W=True, lambda_L=0.5)),
    "ln": dict(n=40, nx=6, ny=7, k=3, m=None, iters=15, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.3, normalize=True)),
    "ls": dict(n=48, nx=8, ny=8, k=3, m=None, iters=40, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.7, tol=1e-3), stop=True),
}


def f14():
    rng = np.random.default_rng(1414)
    out = {}
    for name, c_ in L2FIT.items():
        X, G, W, H = synth(rng, c_["n"], c_["nx"], c_["ny"], c_["k"], c_["m"])
        p = c_["nx"] * c_["ny"]
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((c_["k"], p)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        out[f"{name}_X"], out[f"{name}_W0"], out[f"{name}_H0"] = X, W0, H0
        if G is not None:
            out[f"{name}_G"] = G
        out[f"{name}_shape"] = np.array([c_["nx"], c_["ny"]])
        kw = dict(c_["kw"])
        if not c_.get("stop"):
            kw.update(tol=0, no_stop_criterion=True)
        est = SmoothNMF(n_components=c_["k"], G=G, shape_2d=(c_["nx"], c_["ny"]), verbose=0, algo="l2_surrogate", l2=True,
                        max_iter=c_["iters"], **kw)
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        assert est.l2 is True
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"] = GW, est.W_, est.H_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
        out[f"{name}_n_iter"] = np.array(est.n_iter_)
    out["names"] = np.array(list(L2FIT))
    out["configs"] = np.array(json.dumps(L2FIT))
    save("f14_frobenius_fit", **out)


# ------------------------------------------------------------------ F15: gradients and the Q step (module functions)
def f15():
    from espm.estimators.updates import gradH, gradW, update_q
    rng = np.random.default_rng(1515)
    out = {}
    names = []
    for name, (n, nx, ny, k, m) in {"i": (40, 6, 7, 3, None), "g": (36, 5, 8, 4, 7)}.items():
        X, G, W, H = synth(rng, n, nx, ny, k, m)
        Gd = np.eye(n) if G is None else G
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((k, nx * ny)) + 0.05
        L = create_laplacian_matrix(nx, ny)
        mu = rng.random(k) * 0.3
        out[f"{name}_X"], out[f"{name}_G"], out[f"{name}_W0"], out[f"{name}_H0"], out[f"{name}_mu"] = X, Gd, W0, H0, mu
        out[f"{name}_shape"] = np.array([nx, ny])
        out[f"{name}_gradW"] = gradW(X, Gd, W0, H0)
        out[f"{name}_gradW_l2"] = gradW(X, Gd, W0, H0, l2=True)
        out[f"{name}_gradH"] = gradH(X, Gd, W0, H0, mu=mu, lambda_L=0.8, L=L, epsilon_reg=0.7)
        out[f"{name}_gradH_plain"] = gradH(X, Gd, W0, H0)
        out[f"{name}_gradH_l2"] = gradH(X, Gd, W0, H0, mu=0.2, lambda_L=0.5, L=L, l2=True)
        out[f"{name}_Q"] = update_q(Gd @ W0, H0)
        names.append(name)
    out["names"] = np.array(names)
    save("f15_gradients", **out)


# ------------------------------------------------------------------ F18: the projected-gradient steps with the Frobenius gradient (l2=True)
def f18():
    """espm/estimators/updates.py:353-395 called directly with l2=True (the branch a fit never takes: smooth_nmf.py resets l2 outside
    algo="l2_surrogate"), with and without the simplex over H, fixed entries, a dictionary G."""
    from espm.estimators.updates import proj_grad_step_h, proj_grad_step_w
    rng = np.random.default_rng(1818)
    out = {}
    names = []
    for name, (n, nx, ny, k, m) in {"i": (40, 6, 7, 3, None), "g": (36, 5, 8, 4, 7)}.items():
        X, G, W, H = synth(rng, n, nx, ny, k, m)
        Gd = np.eye(n) if G is None else G
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((k, nx * ny)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        L = create_laplacian_matrix(nx, ny)
        mu = rng.random(k) * 0.3
        fixed_W = np.where(rng.random(W0.shape) < 0.1, rng.random(W0.shape), -1.0)
        fixed_H = np.where(rng.random(H0.shape) < 0.1, 0.5 * rng.random(H0.shape), -1.0)
        gw = 4.0 * float(np.abs(2 * Gd.T @ ((Gd @ W0) @ H0 - X) @ H0.T).max()) / float(W0.mean())   # step sizes that keep most entries off the clamp
        gh = 4.0 * float(np.abs((Gd @ W0).T @ ((Gd @ W0) @ H0 - X)).max())
        out[f"{name}_X"], out[f"{name}_G"], out[f"{name}_W0"], out[f"{name}_H0"], out[f"{name}_mu"] = X, Gd, W0, H0, mu
        out[f"{name}_shape"], out[f"{name}_gamma"] = np.array([nx, ny]), np.array([gh, gw])
        out[f"{name}_fixed_W"], out[f"{name}_fixed_H"] = fixed_W, fixed_H
        out[f"{name}_W_l2"] = proj_grad_step_w(X, Gd, W0, H0, gw, simplex_W=False, l2=True)
        out[f"{name}_W_l2_fixed"] = proj_grad_step_w(X, Gd, W0, H0, gw, simplex_W=False, l2=True, fixed_W=fixed_W)
        out[f"{name}_H_l2"] = proj_grad_step_h(X, Gd, W0, H0, gh, simplex_H=True, l2=True)
        out[f"{name}_H_l2_free"] = proj_grad_step_h(X, Gd, W0, H0, gh, simplex_H=False, mu=mu, lambda_L=0.6, L=L, epsilon_reg=0.8, l2=True, fixed_H=fixed_H)
        out[f"{name}_H_l2_reg"] = proj_grad_step_h(X, Gd, W0, H0, gh, simplex_H=True, mu=0.2, lambda_L=0.5, L=L, l2=True)
        names.append(name)
    out["names"] = np.array(names)
    save("f18_projected_gradient_l2", **out)


def f19():
    """espm/estimators/updates.py:232-261, `multiplicative_step_wq`: the W step "using the WQ technique".  Its docstring says it does exactly
    what `multiplicative_step_w` does; with simplex_W=True (its default) it does not: the multiplier is found for the numerators WITHOUT their
    factor W (dichotomy_simplex(term1, term2), updates.py:253-258), so W' is not on the simplex.  Captured as the reference runs it, beside
    `multiplicative_step_w` on the same inputs; identity G and a dictionary; with the physics model's row subset."""
    from espm.estimators.updates import multiplicative_step_w, multiplicative_step_wq

    class Rows:
        def __init__(self, rows):
            self.rows = rows

        def NMF_simplex(self):
            return self.rows

    rng = np.random.default_rng(1919)
    out = {}
    names = []
    for name, (n, nx, ny, k, m) in {"i": (40, 6, 7, 3, None), "g": (36, 5, 8, 4, 7)}.items():
        X, G, W, H = synth(rng, n, nx, ny, k, m)
        Gd = np.eye(n) if G is None else G
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        H0 = rng.random((k, nx * ny)) + 0.05
        H0 /= H0.sum(axis=0, keepdims=True)
        rows = np.sort(rng.choice(W0.shape[0], size=max(2, W0.shape[0] // 2), replace=False))
        out[f"{name}_X"], out[f"{name}_G"], out[f"{name}_W0"], out[f"{name}_H0"], out[f"{name}_rows"] = X, Gd, W0, H0, rows
        out[f"{name}_wq_simplex"] = multiplicative_step_wq(X, Gd, W0, H0, simplex_W=True)
        out[f"{name}_wq_free"] = multiplicative_step_wq(X, Gd, W0, H0, simplex_W=False)
        out[f"{name}_wq_rows"] = multiplicative_step_wq(X, Gd, W0, H0, simplex_W=True, physics_model=Rows(rows))
        out[f"{name}_w_simplex"] = multiplicative_step_w(X, Gd, W0, H0, simplex_W=True)
        out[f"{name}_w_free"] = multiplicative_step_w(X, Gd, W0, H0, simplex_W=False)
        names.append(name)
    out["names"] = np.array(names)
    save("f19_multiplicative_step_wq", **out)


def f20():
    """The measures of espm/measures.py that the path's own tests lean on and that had no counterpart until round 5 - KL, KLdiv (:387-454),
    KL_loss_surrogate (:506-522), log_surrogate (:550-558), r2 / ordered_r2 (:99-119, :315-327) - and the generic bisection `dicotomy`
    (estimators/dicotomy.py:111-173) on a vector of brackets with a transcendental function."""
    from espm.estimators.dicotomy import dicotomy
    from espm.measures import KL, KL_loss_surrogate, KLdiv, log_surrogate, ordered_r2, r2
    rng = np.random.default_rng(2020)
    n, k, p = 23, 4, 31
    W = rng.random((n, k)) + 0.01
    H = rng.random((k, p)) + 0.01
    Ht = H * (0.5 + rng.random((k, p)))
    X = rng.poisson(40 * W @ H).astype(np.float64) / 40
    X[3, :] = 0
    Y = (W @ H) * (0.7 + 0.6 * rng.random((n, p)))
    mu = rng.random(k) * 0.4
    out = dict(X=X, W=W, H=H, Ht=Ht, Y=Y, mu=mu)
    out["KL"], out["KL_avg"] = KL(X, Y), KL(X, Y, average=True)
    out["KLdiv"], out["KLdiv_avg"] = KLdiv(X, W, H), KLdiv(X, W, H, average=True)
    out["KLs"], out["KLs_avg"], out["KLs_at"] = KL_loss_surrogate(X, W, H, Ht), KL_loss_surrogate(X, W, H, Ht, average=True), KL_loss_surrogate(X, W, Ht, Ht)
    out["logs"], out["logs_avg"], out["logs_scalar"] = log_surrogate(H, Ht, mu, 0.8), log_surrogate(H, Ht, mu, 0.8, average=True), log_surrogate(H, Ht, 0.3, 1.0)
    maps_t = rng.random((k, 6, 7))
    maps_a = maps_t[[2, 0, 3, 1]] + 0.1 * rng.standard_normal((k, 6, 7))
    out["maps_t"], out["maps_a"] = maps_t, maps_a
    out["r2"] = r2(maps_t[0], maps_a[1])
    out["ordered_r2"] = np.array(ordered_r2(maps_t, maps_a, [2, 0, 3, 1]))
    t = rng.random(9) * 3 + 0.2
    lo, hi = np.zeros(9), np.full(9, 8.0)
    out["dic_t"] = t
    out["dic_root"] = dicotomy(lo, hi, lambda x: np.exp(-x) * (t - x) + 0.1 * (t - x), 100, 1e-7)   # (decreasing in x, root x = t)
    out["dic_a"], out["dic_b"] = lo, hi      # (the brackets as the routine leaves them: updated in place)
    save("f20_measures_and_dicotomy", **out)


def f21():
    """The REFERENCE's estimator on the draws of the randomised parity sweep (tests/test_gpu_fuzz.py::_case: every solver and option
    combination, 1..32 components, dictionaries, fixed entries, lines without counts): what the sweep otherwise checks against the oracle
    only.  The inputs are regenerated from the seeds by the tests (the same numpy generator); stored here: whether the reference accepts the
    draw, its losses and the factors it returns (fp32: the comparison is at 1e-4 ... 2e-3)."""
    # (the draws' generator lives with the tests: beside this script's directory, or - when the script was copied elsewhere to check that it
    #  reproduces the committed fixtures - under ESPM_REPO / /root/repo)
    for root in (os.path.dirname(os.path.dirname(HERE)), os.environ.get("ESPM_REPO", ""), "/root/repo"):
        if root and os.path.exists(os.path.join(root, "tests", "test_gpu_fuzz.py")):
            sys.path.insert(0, root)
            sys.path.insert(0, os.path.join(root, "tests"))
            break
    import test_gpu_fuzz as fz
    out = {}
    index = []
    for wide, count in ((0, 24), (1, 12), (2, 12)):
        for seed in range(count):
            c = fz._case(seed, wide=wide)
            tag = f"w{wide}_s{seed}"
            status = "ok"
            try:
                est = SmoothNMF(n_components=c["k"], G=c["G"], shape_2d=c["shape"], algo=c["algo"], tol=0, no_stop_criterion=True, max_iter=6,
                                verbose=0, **c["kw"], **c["extra"])
                GW = quiet(est.fit_transform, c["X"].copy(), W=c["W0"].copy(), H=c["H0"].copy())
                losses = np.array(est.losses_, dtype=float)
                if not np.isfinite(losses).all():
                    status = "nonfinite"
                elif c["algo"] == "projected_gradient" and (np.diff(losses) > 0).any():
                    status = "unstable"
            except AssertionError:
                status = "refused"
            index.append((tag, status))
            if status == "ok":
                out[f"{tag}_losses"] = losses
                out[f"{tag}_W"], out[f"{tag}_H"] = np.asarray(est.W_, dtype=np.float32), np.asarray(est.H_, dtype=np.float32)
                out[f"{tag}_x_sum"] = np.array(float(c["X"].sum()))     # (the test's regenerated input must be this one)
            print(tag, c["algo"], c["k"], status, flush=True)
    out["index"] = np.array(json.dumps(index))
    save("f21_reference_on_the_fuzz_draws", **out)


# ------------------------------------------------------------------ F16: a physics model that refreshes G every third iteration
PHYS = {
    # the reference's default constraint (simplex over the rows NMF_simplex() names), Laplacian
    "pw": dict(n=56, nx=8, ny=9, k=3, m=8, m0=6, iters=13, kw=dict(simplex_H=False, simplex_W=True, lambda_L=0.5)),
    # simplex over H, Laplacian, log sparsity (the C5 analogue with a live model)
    "ph": dict(n=60, nx=10, ny=9, k=4, m=9, m0=7, iters=13, kw=dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0.05)),
    # default stop rules: eval_before is re-evaluated after every refresh (base.py:388-392), which decides n_iter_
    "ps": dict(n=48, nx=8, ny=8, k=3, m=7, m0=5, iters=80, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.3, tol=3e-4), stop=True),
    # Frobenius data term with a live model: the W step divides by G^T G W H H^T of the CURRENT G (updates.py:31-36)
    "pl2": dict(n=44, nx=7, ny=8, k=3, m=6, m0=4, iters=12, kw=dict(simplex_H=True, simplex_W=False, lambda_L=0.6, algo="l2_surrogate", l2=True)),
}


def f16():
    from espm.models.base import PhysicalModel
    sys.path.insert(0, os.path.dirname(HERE))
    from physics_double import AbsorbingModel

    class RefModel(AbsorbingModel, PhysicalModel):   # the reference asks isinstance(G, PhysicalModel) (base.py:269)
        def __init__(self, G0, Abs, strength, m0):
            PhysicalModel.__init__(self, 0.0, G0.shape[0], 0.01, {}, db_name=None)
            AbsorbingModel.__init__(self, G0, Abs, strength, m0)

        def generate_g_matr(self, *a, **k):
            pass

        def generate_phases(self, *a, **k):
            pass

    rng = np.random.default_rng(1616)
    out = {}
    for name, c_ in PHYS.items():
        X, G0, W, H = synth(rng, c_["n"], c_["nx"], c_["ny"], c_["k"], c_["m"])
        p = c_["nx"] * c_["ny"]
        Abs = rng.random((c_["n"], c_["m0"])) * (rng.random((c_["n"], c_["m0"])) < 0.5)
        strength = 0.8
        W0 = rng.random(W.shape) * W.mean() * 2 + 1e-3
        if c_["kw"].get("simplex_W"):
            W0[:c_["m0"]] /= W0[:c_["m0"]].sum(axis=0, keepdims=True)
        H0 = rng.random((c_["k"], p)) + 0.05
        if c_["kw"].get("simplex_H"):
            H0 /= H0.sum(axis=0, keepdims=True)
        else:
            H0 *= X.sum() / (G0 @ W0 @ H0).sum()
        model = RefModel(G0, Abs, strength, c_["m0"])
        kw = dict(c_["kw"])
        if not c_.get("stop"):
            kw.update(tol=0, no_stop_criterion=True)
        est = SmoothNMF(n_components=c_["k"], G=model, shape_2d=(c_["nx"], c_["ny"]), verbose=0, max_iter=c_["iters"], **kw)
        GW = quiet(est.fit_transform, X, W=W0.copy(), H=H0.copy())
        assert est.physics_model_ is model and model.updates in (est.n_iter_ // 3, est.n_iter_ // 3 - 1) and not np.allclose(est.G_, G0)   # (the loop ends before the refresh of its last iteration)
        out[f"{name}_X"], out[f"{name}_G0"], out[f"{name}_Abs"], out[f"{name}_W0"], out[f"{name}_H0"] = X, G0, Abs, W0, H0
        out[f"{name}_strength"] = np.array(strength)
        out[f"{name}_shape"] = np.array([c_["nx"], c_["ny"]])
        out[f"{name}_GW"], out[f"{name}_W"], out[f"{name}_H"], out[f"{name}_G"] = GW, est.W_, est.H_, est.G_
        out[f"{name}_losses"] = np.array(est.losses_)
        out[f"{name}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{name}_rel"] = np.array(est.rel_)
        out[f"{name}_n_iter"] = np.array(est.n_iter_)
        out[f"{name}_updates"] = np.array(model.updates)
    out["names"] = np.array(list(PHYS))
    out["configs"] = np.array(json.dumps(PHYS))
    save("f16_physics_model", **out)


# ------------------------------------------------------------------ F17: the hyperspy calling convention at a size where the
# (pixels, channels) input goes to the device as it lies (espm/estimators/base.py:243-247, :412-420,
# espm/datasets/eds_spim.py:597-604: decomposition(algorithm=est) hands fit_transform the (p, n) matrix)
def f17():
    rng = np.random.default_rng(1717)
    n, nx, ny, k = 512, 96, 96, 3
    p = nx * ny
    # spectra: a few Gaussian lines per phase on a weak continuum; maps: smooth, on the simplex; ~60 counts per pixel
    ch = np.arange(n)[:, None]
    D = np.zeros((n, k))
    for j in range(k):
        for _ in range(4):
            D[:, j] += rng.random() * np.exp(-0.5 * ((ch[:, 0] - rng.integers(20, n - 20)) / (2.0 + 4.0 * rng.random())) ** 2)
        D[:, j] += 0.002 * np.exp(-ch[:, 0] / 200.0)
    D /= D.sum(axis=0, keepdims=True)
    yy, xx = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), indexing="ij")
    maps = np.stack([1.2 + np.cos(3.1 * xx + 0.4) * np.cos(2.3 * yy), 1.2 + np.sin(4.0 * yy + 0.3), 1.2 + np.cos(5.0 * (xx - yy))])
    maps = (maps / maps.sum(axis=0, keepdims=True)).reshape(k, p)
    X = rng.poisson(60.0 * D @ maps).astype(np.float64)                   # (n, p)
    assert X.max() <= 255 and (X.sum(axis=0) > 0).all()
    Xp = np.ascontiguousarray(X.T)                                           # (p, n): what hyperspy hands over
    W0 = rng.random((n, k)) * 60.0 / n + 1e-3
    H0 = rng.random((k, p)) + 0.1
    H0 /= H0.sum(axis=0, keepdims=True)
    out = dict(X_u8=Xp.astype(np.uint8), W0=W0, H0=H0, shape=np.array([nx, ny]))
    for tag, kw, iters in (("free", dict(tol=0, no_stop_criterion=True), 30), ("stop", dict(tol=6e-4), 200)):
        est = SmoothNMF(n_components=k, hspy_comp=True, shape_2d=(nx, ny), simplex_H=True, simplex_W=False, lambda_L=1.0, verbose=0,
                        max_iter=iters, **kw)
        ret = quiet(est.fit_transform, Xp.copy(), W=W0.copy(), H=H0.copy())
        assert ret.shape == (p, k) and est.components_.shape == (k, n)
        out[f"{tag}_loadings"], out[f"{tag}_components"] = ret, est.components_
        out[f"{tag}_W"], out[f"{tag}_H"] = est.W_, est.H_
        out[f"{tag}_losses"], out[f"{tag}_rel"] = np.array(est.losses_), np.array(est.rel_)
        out[f"{tag}_detailed"] = np.array(est.detailed_losses_, dtype=float)
        out[f"{tag}_n_iter"] = np.array(est.n_iter_)
        print(f"f17 {tag}: n_iter {est.n_iter_}, loss {est.losses_[0]:.6g} -> {est.losses_[-1]:.6g}")
    save("f17_hyperspy_ingest", **out)


if __name__ == "__main__":
    todo = {f.__name__: f for f in (f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11, f12, f13, f14, f15, f16, f17, f18, f19, f20, f21)}
    for name in (sys.argv[1:] or list(todo)):   # e.g. `make_golden.py f9` adds a family without rewriting the others
        todo[name]()
