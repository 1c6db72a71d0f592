"""numpy fp64 restatement of espm's SmoothNMF multiplicative-update path (the ORACLE).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Parity pinned by ``tests/golden``.

Every function names the reference lines it restates (paths relative to the reference
repository root, adriente/espm @ v1.1.3).  The code is written from the algorithm, in a
functional style, and keeps the reference's observable quirks:

* the bisection stops on a GLOBAL criterion (max over columns), dicotomy.py:152;
* ``H @ L`` uses the un-clamped input H, updates.py:93-96 precede :104;
* ``G=None`` becomes a dense identity, updates.py:163-166;
* the W-step associates ``(G.T @ R) @ H.T``, updates.py:58-59;
* the KL loss is ``sum(Y) - sum(max(X,eps) * log(Y)) + const``, measures.py:493-504,
  base.py:200-203.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp

# espm/conf.py:55-59
LOG_SHIFT = 1e-14
DICOTOMY_TOL = 1e-5
SIGMA_L = 8
MAXIT_DICHOTOMY = 100


# --------------------------------------------------------------------------------------
# simplex root finder  (espm/estimators/dicotomy.py:4-55 and :111-173)
# --------------------------------------------------------------------------------------
def simplex_bracket(num, den, log_shift=LOG_SHIFT):
    """Bracket [a, b] of the Lagrange multiplier, dicotomy.py:17-49.

    a_j = max over {i : num_ij > 0} of num_ij / 2 - den_ij     (f(a) > 0)
    b_j = 2 * k * max_i num_ij - min_i den_ij                  (f(b) < 0)
    """
    num = np.asarray(num)
    den = np.asarray(den)
    if not (num >= 0).all() or not (den >= 0).all() or not (num.sum(axis=0) > 0).all():
        raise AssertionError("dichotomy_simplex preconditions violated")  # dicotomy.py:17-19
    k = den.shape[0]
    if log_shift > 0 and k * log_shift >= 1:
        raise ValueError("No solution exists!")  # dicotomy.py:22-23
    den_b = np.broadcast_to(den, num.shape)
    cand = np.where(num > 0, num / 2 - den_b, -np.inf)
    a = cand.max(axis=0)
    b = k * num.max(axis=0) / 0.5 - den.min(axis=0)
    b = np.broadcast_to(b, a.shape).copy()
    return a.astype(np.result_type(num, den), copy=True), b


def simplex_residual(nu, num, den, log_shift=LOG_SHIFT):
    """f(nu) = sum_i max(num_i / (nu + den_i), eps) - 1, dicotomy.py:51-53."""
    return np.maximum(num / (nu + den), log_shift).sum(axis=0) - 1


def bisect(a, b, func, maxit, tol):
    """Vectorised bisection with the reference's GLOBAL stop rule, dicotomy.py:111-173.

    ``a`` / ``b`` are updated in place exactly like the reference does.
    Returns (root estimate, number of sweeps).
    """
    fa0, fb0 = func(a), func(b)
    if (fb0 >= 0).any() or (fa0 <= 0).any() or np.isnan(fa0).any() or np.isnan(fb0).any():
        raise AssertionError("bisection bracket is not a sign change")  # dicotomy.py:141-144
    sweeps = 0
    mid = (a + b) / 2
    fmid = func(mid)
    while np.max(np.abs(fmid)) > tol:
        sweeps += 1
        to_b = func(a) * fmid <= 0  # dicotomy.py:155-159
        b[to_b] = mid[to_b]
        a[~to_b] = mid[~to_b]
        mid = (a + b) / 2
        fmid = func(mid)
        if sweeps >= maxit:
            break
    return mid, sweeps


def dichotomy_simplex(num, denum, log_shift=LOG_SHIFT, tol=DICOTOMY_TOL, maxit=MAXIT_DICHOTOMY,
                      return_sweeps=False):
    """nu (p,) such that sum_i max(num_ij/(nu_j+den_ij), eps) = 1, dicotomy.py:4-55."""
    num = np.asarray(num)
    denum = np.asarray(denum)
    a, b = simplex_bracket(num, denum, log_shift)
    nu, sweeps = bisect(a, b, lambda x: simplex_residual(x, num, denum, log_shift), maxit, tol)
    return (nu, sweeps) if return_sweeps else nu


def dichotomy_simplex_exact(num, den, log_shift=LOG_SHIFT, sweeps=200):
    """The same root, converged and well conditioned (NOT the reference's arithmetic).

    The reference bisects on nu itself; when the simplex pushes mass onto a component whose
    numerator is tiny (an NNDSVD zero clamped to 1e-14) the root sits ~1e-11 to the right of the
    pole nu = -den_i and fp64 resolves nu + den_i only to ulp(den_i): the reference's column sums
    are then off by up to ~1e-2 (measured 5e-3 on tests/golden/f8_api norm_X).  Here the unknown is
    delta = nu + min{den_i : num_i > 0}, which has full relative precision.  Used by the GPU tests as
    the yardstick in exactly those cases.  Returns (delta, shifted denominators e) with
    update = num / (delta + e)."""
    num = np.asarray(num, dtype=np.float64)
    den = np.broadcast_to(np.asarray(den, dtype=np.float64), num.shape)
    k = num.shape[0]
    dstar = np.where(num > 0, den, np.inf).min(axis=0)
    e = den - dstar
    lo = np.where(num > 0, num / 2 - e, -np.inf).max(axis=0)
    hi = 2 * k * num.max(axis=0) - den.min(axis=0) + dstar

    def f(x):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(num > 0, num / (x + e), 0.0)
        return np.maximum(t, log_shift).sum(axis=0) - 1

    for _ in range(sweeps):
        mid = (lo + hi) / 2
        pos = f(mid) > 0
        lo = np.where(pos, mid, lo)
        hi = np.where(pos, hi, mid)
    return (lo + hi) / 2, e


# --------------------------------------------------------------------------------------
# Laplacian  (espm/utils.py:39-76)
# --------------------------------------------------------------------------------------
def laplacian_matrix(nx, ny=None):
    """Sparse (p, p) 5-point graph Laplacian, zero-flux boundary, row-major pixel index.

    Entry (q, q) is the number of in-image 4-neighbours of pixel q, entry (q, r) is -1 for
    each neighbour r (utils.py:56-75).  float32 like the reference.
    """
    if ny is None:
        ny = nx
    assert nx > 1 and ny > 1  # utils.py:58-59
    idx = np.arange(nx * ny).reshape(nx, ny)
    rows, cols = [], []
    for src, dst in ((idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])):
        rows += [src.ravel(), dst.ravel()]
        cols += [dst.ravel(), src.ravel()]
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    adj = sp.coo_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(nx * ny,) * 2).tocsr()
    deg = sp.diags(np.asarray(adj.sum(axis=1)).ravel().astype(np.float32))
    return (deg - adj).tocsr()


def laplacian_apply(H, nx, ny):
    """(H @ L) for the matrix above, written as a stencil on the (nx, ny) grid."""
    k = H.shape[0]
    img = H.reshape(k, nx, ny)
    out = np.zeros_like(img)
    out[:, 1:, :] += img[:, 1:, :] - img[:, :-1, :]
    out[:, :-1, :] += img[:, :-1, :] - img[:, 1:, :]
    out[:, :, 1:] += img[:, :, 1:] - img[:, :, :-1]
    out[:, :, :-1] += img[:, :, :-1] - img[:, :, 1:]
    return out.reshape(k, nx * ny)


def identity_L(p):
    """L_ when shape_2d is None, base.py:289-291."""
    return sp.identity(p, dtype=np.float32, format="csr")


# --------------------------------------------------------------------------------------
# update rules  (espm/estimators/updates.py:6-78, :83-156)
# --------------------------------------------------------------------------------------
def multiplicative_step_h(X, G, W, H, simplex_H=False, mu=0, log_shift=LOG_SHIFT, epsilon_reg=1,
                          safe=True, dicotomy_tol=DICOTOMY_TOL, lambda_L=0, L=None, l2=False,
                          sigmaL=SIGMA_L, fixed_H=None, exact_root=False, use_bregman=False):
    """One multiplicative H update, updates.py:83-156 (KL branch :127-132, l2 branch :109-118, Bregman variant
    :120-125).

    ``exact_root`` swaps the reference's bisection for ``dichotomy_simplex_exact`` (test yardstick)."""
    if lambda_L != 0:
        if L is None:
            raise ValueError("Please provide the laplacian")  # updates.py:94-95
        HL = H @ L  # from the un-clamped H
    if safe:  # updates.py:98-105
        assert (H >= -log_shift / 2).all() and (W >= -log_shift / 2).all() and (G >= -log_shift / 2).all()
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    GW = G @ W
    if l2:
        assert lambda_L == 0 and np.all(np.asarray(mu) == 0)
        num = GW.T @ X
        den = (GW.T @ GW) @ H
    else:
        if use_bregman:  # updates.py:120-125
            sigmaR = X.sum(axis=0, keepdims=True)
            num = sigmaR / H
            den = -GW.T @ (X / (GW @ H)) + GW.sum(axis=0)[:, None] + sigmaR / H
        else:
            Y = GW @ H
            num = GW.T @ (X / Y)
            if np.isnan(num).any():  # updates.py:129-131
                num = GW.T @ (X / np.maximum(Y, log_shift))
            den = GW.sum(axis=0)[:, None]
        if not (np.isscalar(mu) and mu == 0):
            mu_col = np.asarray(mu, dtype=float)
            if mu_col.ndim == 1:
                mu_col = mu_col[:, None]
            den = den + mu_col / (H + epsilon_reg)  # updates.py:134-137
        if lambda_L != 0:
            maxH = H.max(axis=1, keepdims=True)  # GLOBAL over pixels, updates.py:139
            num = num + lambda_L * sigmaL * maxH
            den = den + lambda_L * sigmaL * maxH + lambda_L * HL
    num = H * num
    if safe:
        assert (den >= 0).all() and (num >= 0).all()
    if simplex_H and exact_root:
        delta, e = dichotomy_simplex_exact(num, den, log_shift)
        with np.errstate(divide="ignore", invalid="ignore"):
            new_H = np.fmax(num / (delta + e), log_shift)
    else:
        nu = dichotomy_simplex(num, den, log_shift=log_shift, tol=dicotomy_tol) if simplex_H else 0
        new_H = np.maximum(num / (den + nu), log_shift)
    if fixed_H is not None:
        keep = fixed_H >= 0
        new_H[keep] = fixed_H[keep]
    return new_H


def multiplicative_step_w(X, G, W, H, simplex_W=False, log_shift=LOG_SHIFT, safe=True, l2=False,
                          fixed_W=None, simplex_rows=None, use_bregman=False):
    """One multiplicative W update, updates.py:6-78 (Bregman variant :40-48: no simplex over W there).

    ``simplex_rows`` stands for ``physics_model.NMF_simplex()`` (updates.py:62-65).
    """
    if safe:
        assert (H >= -log_shift / 2).all() and (W >= -log_shift / 2).all() and (G >= -log_shift / 2).all()
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    if l2:
        new_W = W / ((G.T @ G) @ W @ (H @ H.T)) * (G.T @ (X @ H.T))  # updates.py:30-36
    elif use_bregman:
        # updates.py:41-48 (np.allclose(G, np.eye(n)) needs a square G: the reference fails for a dictionary)
        sigmaR = X.sum(axis=1, keepdims=True) if np.allclose(G, np.eye(G.shape[0])) else np.sum(X)
        gradg = -G.T @ (X / ((G @ W) @ H)) @ H.T + G.sum(axis=0)[:, None] @ H.sum(axis=1)[None, :]
        new_W = (sigmaR * W) / (gradg * W + sigmaR)
    else:
        Y = (G @ W) @ H
        R = X / Y
        if np.isnan(R).any():  # updates.py:54-56
            R = X / np.maximum(Y, log_shift)
        num = W * ((G.T @ R) @ H.T)  # the reference's association
        den = G.sum(axis=0)[:, None] @ H.sum(axis=1)[None, :]
        if simplex_W:
            if simplex_rows is not None:
                nu = dichotomy_simplex(num[simplex_rows, :], den[simplex_rows, :], log_shift=log_shift,
                                       tol=DICOTOMY_TOL)
                den[simplex_rows, :] = den[simplex_rows, :] + nu
            else:
                den = den + dichotomy_simplex(num, den, log_shift=log_shift, tol=DICOTOMY_TOL)
        new_W = num / den
    new_W = np.maximum(new_W, log_shift)
    if fixed_W is not None:
        keep = fixed_W >= 0
        new_W[keep] = fixed_W[keep]
    return new_W


# --------------------------------------------------------------------------------------
# losses  (espm/measures.py:456-504, :524-548, :560-577; base.py:167-207; smooth_nmf.py:457-475)
# --------------------------------------------------------------------------------------
def KLdiv_loss(X, D, H, log_shift=LOG_SHIFT, average=False):
    """sum(Y) - sum(max(X,eps) log Y) with clamped D, H, measures.py:456-504."""
    Y = np.maximum(D, log_shift) @ np.maximum(H, log_shift)
    Xc = np.maximum(X, log_shift)
    red = np.mean if average else np.sum
    return red(Y) - red(Xc * np.log(Y))


def const_KL(X, log_shift=LOG_SHIFT):
    """base.py:200-201."""
    return np.sum(X * np.log(np.maximum(X, log_shift))) - np.sum(X)


def log_reg(H, mu, epsilon=1, average=False):
    """sum_ij mu_i log(H_ij + eps), measures.py:524-548."""
    mu = mu if np.isscalar(mu) else np.asarray(mu)[:, None]
    red = np.mean if average else np.sum
    return red(mu * np.log(H + epsilon))


def trace_xtLx(L, x, average=False):
    """sum(x * (L @ x)), measures.py:560-577."""
    red = np.mean if average else np.sum
    return red(x * (L @ x))


def Frobenius_loss(X, D, H, average=False):
    """measures.py:350-385."""
    red = np.mean if average else np.sum
    return red((D @ H - X) ** 2)


def smooth_nmf_loss(X, G, W, H, L, mu=0, epsilon_reg=1, lambda_L=0.0, log_shift=LOG_SHIFT,
                    average=True, c_kl=None, gamma=SIGMA_L, l2=False):
    """SmoothNMF.loss: returns (total, [lkl, reg, lap, gamma]), smooth_nmf.py:457-475; data term base.py:197-203
    (l2: half the squared Frobenius distance instead of the KL divergence)."""
    numel = G.shape[0] * H.shape[1]
    if l2:
        lkl = 0.5 * Frobenius_loss(X, G @ W, H, average=False)
    else:
        if c_kl is None:
            c_kl = const_KL(X, log_shift)
        lkl = KLdiv_loss(X, G @ W, H, log_shift) + c_kl
    reg = log_reg(H, mu, epsilon_reg)
    lap = 0.5 * lambda_L * trace_xtLx(L, H.T)
    if average:
        lkl, reg, lap = lkl / numel, reg / numel, lap / numel
    return lkl + reg + lap, [lkl, reg, lap, gamma]


# --------------------------------------------------------------------------------------
# initialisation and fit loop  (updates.py:160-223, base.py:16-18, :209-420, :519-528)
# --------------------------------------------------------------------------------------
def normalization_factor(X, nc):
    """base.py:16-18."""
    return nc / (np.mean(X) * X.shape[0])


def remove_zeros_lines(X, epsilon):
    """All-zero rows / columns of X become epsilon, base.py:519-528."""
    if not np.all(X >= 0):
        raise ValueError("Negative values in data")
    out = X.copy()
    out[:, X.sum(axis=0) == 0] = epsilon
    out[X.sum(axis=1) == 0, :] = epsilon
    return out


def initialize_algorithms(X, G, W, H, n_components, init, random_state, simplex_H, simplex_W,
                          logshift=LOG_SHIFT):
    """updates.py:160-223 without the physics-model branch (G is None or an ndarray)."""
    from sklearn.decomposition._nmf import _initialize_nmf

    identity = G is None
    if identity:
        G = np.diag(np.ones(X.shape[0]).astype(X.dtype))
    if W is None:
        if H is None:
            D, H = _initialize_nmf(X, n_components=n_components, init=init, random_state=random_state)
            if simplex_H:
                H = np.nan_to_num(H, nan=1.0 / H.shape[0])
                scale = H.sum(axis=0, keepdims=True)
                H = H / scale
                D = D * np.mean(scale)
        else:
            D = np.abs(np.linalg.lstsq(H.T, X.T, rcond=None)[0].T)
        if identity:
            W = D
        else:
            W = np.abs(np.linalg.lstsq(G, D, rcond=None)[0])
            if simplex_W:
                W = np.nan_to_num(W, nan=1.0 / W.shape[0])
                W = W / W.sum(axis=0, keepdims=True)
    elif H is None:
        H = np.abs(np.linalg.lstsq(G @ W, X, rcond=None)[0])
        if simplex_H:
            H = H / H.sum(axis=0, keepdims=True)
    return G, np.maximum(W, logshift), np.maximum(H, logshift)


def rescaled_DH(D, H):
    """espm/utils.py:79-96."""
    from scipy.optimize import nnls

    o = np.ones(H.shape[1])
    s = np.linalg.lstsq(H.T, o, rcond=None)[0]
    if (s <= 0).any():
        s = np.maximum(nnls(H.T, o)[0], 1e-10)
    return D @ np.diag(1 / s), np.diag(s) @ H


# ---- quadratic ("l2") surrogate of the Laplacian term: algo = "l2_surrogate"  -------------------
def dichotomy_simplex_acc(a, b, minus_c, log_shift=LOG_SHIFT, tol=DICOTOMY_TOL, maxit=MAXIT_DICHOTOMY):
    """nu (p,) with sum_k max(sqrt((b_kj + nu_j)^2 + 4 a c_kj) - nu_j - b_kj, 2 a eps) = 2 a, dicotomy.py:57-82."""
    assert a >= 0 and (minus_c >= 0).all()
    if log_shift > 0 and b.shape[0] * log_shift >= 1:
        raise ValueError("No solution exists!")
    n_p = len(b)
    nu_max = n_p * np.max(b ** 2 / a + 2 * a + 2 * (b + minus_c), axis=0) * 1.5 + 1e-3
    nu_min = -(2 * a + np.sum(b, axis=0)) / n_p * 1.1 - 1e-3

    def func(x):
        return 2 * a - np.sum(np.maximum(np.sqrt((b + x) ** 2 + 4 * a * minus_c) - x - b, log_shift * 2 * a), axis=0)

    return bisect(nu_max, nu_min, func, maxit, tol)[0]


def multiplicative_step_hq(X, G, W, H, simplex_H=True, log_shift=LOG_SHIFT, safe=True, dicotomy_tol=DICOTOMY_TOL,
                           lambda_L=0, L=None, sigmaL=SIGMA_L, fixed_H=None):
    """updates.py:263-315: H update from the quadratic surrogate of the Laplacian term - the positive root of
    a H'^2 + b H' - c = 0 with a = lambda sigma, b = colsum(GW) + lambda (H L) - lambda sigma H (+ nu), c = H GW^T (X / GWH)."""
    if lambda_L != 0 and L is None:
        raise ValueError("Please provide the laplacian")
    if safe:
        assert (H >= -log_shift / 2).all() and (W >= -log_shift / 2).all() and (G >= -log_shift / 2).all()
    GW = G @ W
    minus_c = H * (GW.T @ (X / (GW @ H + log_shift)))
    b = GW.sum(axis=0)[:, None]
    if lambda_L != 0:
        b = b + lambda_L * (H @ L) - lambda_L * sigmaL * H
        a = lambda_L * sigmaL
        if simplex_H:
            b = b + dichotomy_simplex_acc(a, b, minus_c, log_shift=log_shift, tol=dicotomy_tol)
        new_H = (-b + np.sqrt(b ** 2 + 4 * a * minus_c)) / (2 * a)
    else:  # the classic case
        if simplex_H:
            b = b + dichotomy_simplex(minus_c, b, log_shift=log_shift, tol=dicotomy_tol)
        new_H = minus_c / b
    new_H = np.maximum(new_H, log_shift)
    if fixed_H is not None:
        keep = fixed_H >= 0
        new_H[keep] = fixed_H[keep]
    return new_H


def smooth_l2_surrogate(Ht, L, H, sigmaL=SIGMA_L, lambda_L=1):
    """espm/estimators/surrogates.py:6-58: lambda/2 (2 tr(Ht L H^T) - tr(Ht L Ht^T) + sigma ||Ht - H||^2)."""
    HtL = Ht @ L
    return lambda_L / 2 * (2 * np.sum(HtL * H) - np.sum(HtL * Ht) + sigmaL * np.sum((Ht - H) ** 2))


# ---- projected gradient (algo = "projected_gradient"; without its linesearch) ------------------------
def update_q(D, H, log_shift=LOG_SHIFT):
    """updates.py:225-230: Q[i, j, k] = H[k, j] D[i, k] / ((D H)[i, j] + log_shift)."""
    return H.T[None, :, :] * (D[:, None, :] / ((D @ H)[:, :, None] + log_shift))


def multiplicative_step_wq(X, G, W, H, simplex_W=True, log_shift=LOG_SHIFT, safe=True, rows=None):
    """updates.py:232-261, the W step "using the WQ technique", as the reference runs it: term1 = G^T (XQ / (GW + log_shift)) with
    XQ = sum_j X[:, j] Q[:, j, :], term2 = colsum(G)^T rowsum(H)^T, W' = W / (term2 + nu) * term1 - where nu comes from
    dichotomy_simplex(term1, term2), the numerators WITHOUT their factor W (updates.py:253-258): with simplex_W the result is not on the
    simplex, whatever the docstring says.  rows: the physics model's NMF_simplex() subset."""
    if safe:
        assert np.sum(H < -log_shift / 2) == 0 and np.sum(W < -log_shift / 2) == 0 and np.sum(G < -log_shift / 2) == 0
    GW = G @ W
    Q = update_q(GW, H, log_shift=log_shift)
    XQ = np.sum(X[:, :, None] * Q, axis=1)
    term1 = G.T @ (XQ / (GW + log_shift))
    term2 = np.sum(G, axis=0, keepdims=True).T @ np.sum(H, axis=1, keepdims=True).T
    if simplex_W:
        if rows is not None:
            nu = dichotomy_simplex(term1[rows, :], term2[rows, :], log_shift=log_shift, tol=DICOTOMY_TOL)
            term2[rows, :] = term2[rows, :] + nu
        else:
            term2 = term2 + dichotomy_simplex(term1, term2, log_shift=log_shift, tol=DICOTOMY_TOL)
    return W / term2 * term1


def gradW(X, G, W, H, log_shift=LOG_SHIFT, safe=False, l2=False):
    """updates.py:303-313: G^T (-(X / GWH) H^T + rowsum(H)^T); l2: 2 G^T (GWH - X) H^T."""
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    if l2:
        return 2 * G.T @ ((G @ W) @ H - X) @ H.T
    return G.T @ (-(X / ((G @ W) @ H)) @ H.T + np.sum(H, axis=1, keepdims=True).T)


def gradH(X, G, W, H, mu=0, lambda_L=0, L=None, epsilon_reg=1, log_shift=LOG_SHIFT, safe=False, l2=False):
    """updates.py:315-342: -GW^T (X / GWH) + colsum(GW) (l2: GW^T (GWH - X)) + mu / (H + eps) + lambda (L H^T)^T."""
    if lambda_L != 0 and L is None:
        raise ValueError("Please provide the laplacian")
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    D = G @ W
    grad = D.T @ (D @ H - X) if l2 else -D.T @ (X / (D @ H)) + np.sum(D, axis=0, keepdims=True).T
    if not (np.isscalar(mu) and mu == 0):
        mu_col = np.asarray(mu, dtype=float)
        if mu_col.ndim == 1:
            mu_col = mu_col[:, None]
        grad = grad + mu_col / (H + epsilon_reg)
    if lambda_L != 0:
        grad = grad + (lambda_L * (L @ H.T)).T
    return grad


def dichotomy_simplex_projected_gradient(a, log_shift=LOG_SHIFT, tol=DICOTOMY_TOL, maxit=MAXIT_DICHOTOMY):
    """nu (p,) with sum_k max(a_kj + nu_j, eps) = 1, dicotomy.py:84-108."""
    if log_shift > 0 and a.shape[0] * log_shift >= 1:
        raise ValueError("No solution exists!")
    nu_min = -np.max(a, axis=0)
    nu_max = 1 / a.shape[0] - np.min(a, axis=0)
    return bisect(nu_max, nu_min, lambda x: np.sum(np.maximum(a + x, log_shift), axis=0) - 1, maxit, tol)[0]


def proj_grad_step_w(X, G, W, H, gamma, simplex_W=True, log_shift=LOG_SHIFT, safe=True, fixed_W=None, l2=False):
    """updates.py:353-370: W - grad / gamma, clamped; no simplex over W with this method (l2: the Frobenius gradient, :361)."""
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    new_W = np.maximum(W - 1 / gamma * gradW(X, G, W, H, log_shift=log_shift, safe=safe, l2=l2), log_shift)
    if fixed_W is not None:
        keep = fixed_W >= 0
        new_W[keep] = fixed_W[keep]
    if simplex_W:
        raise NotImplementedError("Simplex constraint not implemented for W using the projected gradient method")
    return new_W


def proj_grad_step_h(X, G, W, H, gamma, simplex_H=True, mu=0, log_shift=LOG_SHIFT, epsilon_reg=1, safe=True,
                     dicotomy_tol=DICOTOMY_TOL, lambda_L=0, L=None, fixed_H=None, l2=False):
    """updates.py:372-395: H - grad / gamma, projected on the simplex (or just clamped); l2: the Frobenius gradient (:380)."""
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    new_H = H - 1 / gamma * gradH(X, G, W, H, log_shift=log_shift, safe=safe, mu=mu, epsilon_reg=epsilon_reg, lambda_L=lambda_L, L=L, l2=l2)
    nu = dichotomy_simplex_projected_gradient(new_H, log_shift=log_shift, tol=dicotomy_tol) if simplex_H else 0
    new_H = np.maximum(new_H + nu, log_shift)
    if fixed_H is not None:
        keep = fixed_H >= 0
        new_H[keep] = fixed_H[keep]
    return new_H


def quadratic_surrogate(x, xt, f_xt, gradf_xt, sigma):
    """espm/estimators/surrogates.py:153-170: f(xt) + <x - xt, grad f(xt)> + sigma ||x - xt||^2."""
    return f_xt + np.sum((x - xt) * gradf_xt) + sigma * np.sum((x - xt) ** 2)


# ---- linesearch on the Laplacian surrogate -------------------------------------------
def smooth_dgkl_surrogate(Ht, L, H, sigmaL=SIGMA_L, lambda_L=1):
    """espm/estimators/surrogates.py:65-114: lambda/2 (2 tr(Ht L H^T) - tr(Ht L Ht^T) + sigma sum_k max_j H_kj sum_j dgkl(Ht_kj, H_kj))."""
    HtL = Ht @ L
    t1 = np.sum(HtL * Ht)
    t2 = np.sum(HtL * H)
    t3 = np.sum(np.max(H, axis=1) * np.sum(Ht * np.log(Ht / H) - Ht + H, axis=1))
    return lambda_L / 2 * (2 * t2 - t1 + sigmaL * t3)


def diff_surrogate(Ht, H, L, sigmaL=SIGMA_L, lambda_L=1, algo="log_surrogate"):
    """espm/estimators/surrogates.py:116-149: surrogate ("log_surrogate" / "bmd": dgkl; "l2_surrogate": quadratic)
    minus the Laplacian term at H."""
    surr = smooth_l2_surrogate if algo == "l2_surrogate" else smooth_dgkl_surrogate
    return surr(Ht, L, H, sigmaL, lambda_L) - trace_xtLx(L, H.T) * lambda_L / 2


def linesearch_gamma(gamma, Hold, H, L, algo="log_surrogate"):
    """espm/estimators/smooth_nmf.py:376-381: the caller passes neither lambda_L nor its own value - diff_surrogate runs
    with its default lambda_L = 1 whatever the estimator's lambda_L is."""
    return gamma / 1.05 if diff_surrogate(Hold, H, L, sigmaL=gamma, algo=algo) > 0 else gamma * 1.5


# ---- truth tracking (true_D / true_H; base.py:301-347, measures.py) ---------------------------------------
def spectral_angle(v1, v2):
    """espm/measures.py:13-47 (2-D branch): angles in degrees between the rows of v1 and the rows of v2."""
    a = v1 / np.sqrt(np.sum(v1 ** 2, axis=1, keepdims=True))
    b = v2 / np.sqrt(np.sum(v2 ** 2, axis=1, keepdims=True))
    return np.arccos(np.clip(a @ b.T, -1.0, 1.0)) * 180 / np.pi


def squared_distance(x, y):
    """espm/measures.py:579-626: mean squared distance between the rows of x and the rows of y."""
    xx = (x * x).sum(axis=1)
    yy = (y * y).sum(axis=1)
    return np.abs(xx[:, None] + yy[None, :] - 2 * (x @ y.T)) / x.shape[1]


def unique_min(matrix):
    """espm/measures.py:172-207: the assignment (one row per column) with the smallest sum, by brute force over the
    permutations in itertools order (the first minimum wins); returns its entries in column order."""
    from itertools import permutations
    k = matrix.shape[0]
    perms = list(permutations(range(k), k))
    sums = [sum(matrix[perm[i], i] for i in range(k)) for perm in perms]   # same order of additions as the reference
    best = perms[sums.index(min(sums))]
    return [matrix[best[i], i] for i in range(k)], best


def truth_metrics(true_D, true_H, GW, H):
    """base.py:342-343: find_min_angle(true_D.T, GW.T, unique=True), find_min_MSE(true_H, H, unique=True)."""
    return unique_min(spectral_angle(true_D.T, GW.T))[0], unique_min(squared_distance(true_H, H))[0]


def fit(X, n_components, G=None, W=None, H=None, *, lambda_L=0.0, mu=0, epsilon_reg=1,
        simplex_H=False, simplex_W=True, shape_2d=None, tol=1e-4, max_iter=200, init=None,
        random_state=None, normalize=False, log_shift=LOG_SHIFT, dicotomy_tol=DICOTOMY_TOL,
        gamma=None, fixed_H=None, fixed_W=None, no_stop_criterion=False, safe=False,
        record_at=(), time_iterations=False, exact_root=False, linesearch=False, true_D=None, true_H=None,
        algo="log_surrogate", l2=False, physics_model=None):
    """Reference-faithful fit loop: NMFEstimator.fit_transform (base.py:209-420) driving
    SmoothNMF._iteration (smooth_nmf.py:284-455, algo="log_surrogate", "bmd", "l2_surrogate" or "projected_gradient" with a given gamma; linesearch: smooth_nmf.py:376-381;
    true_D / true_H tracking: base.py:301-347).

    ``physics_model`` (base.py:269-274, :388-392; interface espm/models/base.py:217-264): an object with ``NMF_update(W=None)
    -> G``, ``NMF_simplex() -> rows under the simplex over W`` and ``NMF_initialize_W(D)``; G comes from it, is refreshed
    from the current W after every third iteration, and the loss is re-evaluated with the new G before the next stop test.

    Returns a dict with W, H, G, GW, losses, detailed_losses, rel, n_iter, exit, snapshots.
    """
    if physics_model is not None:
        G = physics_model.NMF_update()
    simplex_rows = physics_model.NMF_simplex() if physics_model is not None else None
    if simplex_H and simplex_W:  # smooth_nmf.py:218-222
        simplex_W, simplex_H = True, False
    X_ = remove_zeros_lines(np.asarray(X), log_shift)
    norm = None
    if normalize:
        norm = normalization_factor(X_, n_components)
        X_ = norm * X_
    G_, W_, H_ = initialize_algorithms(X_, G, W, H, n_components, init, random_state, simplex_H, simplex_W)
    p = X_.shape[1]
    L_ = laplacian_matrix(*shape_2d) if shape_2d is not None else identity_L(p)
    gamma_ = SIGMA_L if gamma is None else gamma  # smooth_nmf.py:290-306
    if algo not in ("log_surrogate", "bmd", "l2_surrogate", "projected_gradient"):
        raise NotImplementedError(algo)
    if algo == "projected_gradient":  # smooth_nmf.py:297-306: a list [gamma_H, gamma_W] (the Lipschitz default is not restated)
        if gamma is None:
            raise NotImplementedError("projected_gradient: pass gamma=[gamma_H, gamma_W]")
        gamma_ = list(gamma)
    l2 = bool(l2) and algo == "l2_surrogate" and not linesearch  # smooth_nmf.py:223-237: elsewhere the flag is switched off
    # (with it, the W step takes its Frobenius branch - no simplex, updates.py:31-36 - and the loss its Frobenius data
    # term, base.py:197-198; the H step of "l2_surrogate" has no such branch, smooth_nmf.py:311-323)
    breg = algo == "bmd"  # smooth_nmf.py:358-372, :416-426: both steps with use_bregman=True
    c_kl = const_KL(X_, log_shift)

    def loss(Wc, Hc, Xc=None):
        # base.py:196-203: with another X (the noiseless truth) the constant cached for the DATA is still the one added
        return smooth_nmf_loss(X_ if Xc is None else Xc, G_, Wc, Hc, L_, mu, epsilon_reg, lambda_L, log_shift, True, c_kl,
                               gamma_[0] if isinstance(gamma_, list) else gamma_, l2=l2)   # smooth_nmf.py:470-473

    def loss_sum(Wc, Hc):
        return smooth_nmf_loss(X_, G_, Wc, Hc, L_, mu, epsilon_reg, lambda_L, log_shift, False, c_kl)[0]

    gammas = []
    track = true_D is not None and true_H is not None and true_D.shape[1] == n_components and true_H.shape[0] == n_components
    true_DH = true_D @ true_H if track else None
    angles, mses, true_losses = [], [], []

    eval_before = np.inf
    # the reference evaluates the initial loss before gamma_ is set (it is None then); only the
    # value matters here
    eval_init, _ = loss(W_, H_)
    losses, detailed, rel, snaps = [], [], [], {}
    n_iter, reason = 0, None
    t0 = time.perf_counter()
    while True:
        old_W, old_H = W_.copy(), H_.copy()
        if algo == "projected_gradient":  # smooth_nmf.py:340-353
            H_ = proj_grad_step_h(X_, G_, W_, H_, gamma_[0], simplex_H=simplex_H, mu=mu, log_shift=log_shift,
                                  epsilon_reg=epsilon_reg, safe=safe, dicotomy_tol=dicotomy_tol, lambda_L=lambda_L, L=L_,
                                  fixed_H=fixed_H)
        elif algo == "l2_surrogate":  # smooth_nmf.py:311-323 (mu does not enter this update)
            H_ = multiplicative_step_hq(X_, G_, W_, H_, simplex_H=simplex_H, log_shift=log_shift, safe=safe,
                                        dicotomy_tol=dicotomy_tol, lambda_L=lambda_L, L=L_, sigmaL=gamma_, fixed_H=fixed_H)
        else:
            H_ = multiplicative_step_h(X_, G_, W_, H_, simplex_H=simplex_H, mu=mu, log_shift=log_shift,
                                       epsilon_reg=epsilon_reg, safe=safe, dicotomy_tol=dicotomy_tol,
                                       lambda_L=lambda_L, L=L_, l2=False, fixed_H=fixed_H, sigmaL=gamma_,
                                       exact_root=exact_root, use_bregman=breg)
        if linesearch and algo == "projected_gradient":  # smooth_nmf.py:382-401 (losses not averaged)
            grad = gradH(X_, G_, W_, old_H, mu=mu, lambda_L=lambda_L, L=L_, epsilon_reg=epsilon_reg, log_shift=log_shift, safe=safe)
            f_xt, f_x = loss_sum(W_, old_H), loss_sum(W_, H_)
            gamma_[0] = gamma_[0] / 1.05 if quadratic_surrogate(H_, old_H, f_xt, grad, gamma_[0]) - f_x > 0 else gamma_[0] * 1.5
        elif linesearch:
            gamma_ = linesearch_gamma(gamma_, old_H, H_, L_, algo)
        if algo == "projected_gradient":  # smooth_nmf.py:427-447 (fixed_W is not passed there)
            W_ = proj_grad_step_w(X_, G_, W_, H_, gamma_[1], simplex_W=simplex_W, log_shift=log_shift, safe=safe)
            if linesearch:
                grad = gradW(X_, G_, old_W, H_, log_shift=log_shift, safe=safe)
                f_xt, f_x = loss_sum(old_W, H_), loss_sum(W_, H_)
                gamma_[1] = gamma_[1] / 1.05 if quadratic_surrogate(W_, old_W, f_xt, grad, gamma_[1]) - f_x > 0 else gamma_[1] * 1.5
        else:
            W_ = multiplicative_step_w(X_, G_, W_, H_, log_shift=log_shift, safe=safe, l2=l2,
                                       simplex_W=simplex_W, fixed_W=fixed_W, use_bregman=breg, simplex_rows=simplex_rows)
        eval_after, det = loss(W_, H_)
        n_iter += 1
        gammas.append(list(gamma_) if isinstance(gamma_, list) else gamma_)
        if track:  # base.py:335-347
            Wt, Ht = (W_, H_) if (simplex_H or simplex_W) else rescaled_DH(W_, H_)
            a_, m_ = truth_metrics(true_D, true_H, G_ @ Wt, Ht)
            angles.append(a_)
            mses.append(m_)
            true_losses.append(loss(W_, Ht, true_DH)[0])
        rel_W = np.max(np.abs(W_ - old_W) / (W_ + tol * np.mean(W_)))  # base.py:323-324
        rel_H = np.max(np.abs(H_ - old_H) / (H_ + tol * np.mean(H_)))
        losses.append(eval_after)
        detailed.append(det)
        rel.append([rel_W, rel_H])
        if n_iter in record_at:
            snaps[n_iter] = (W_.copy(), H_.copy())
        if n_iter >= max_iter:  # base.py:354-378
            reason = "max_iter"
            break
        if not no_stop_criterion:
            if max(rel_H, rel_W) < tol:
                reason = "rel"
                break
            elif abs((eval_before - eval_after) / eval_init) < tol:
                reason = "loss"
                break
            elif np.isnan(eval_after):
                reason = "nan"
                break
            elif (eval_before - eval_after) < 0:
                reason = "increase"
                break
        if physics_model is not None and n_iter % 3 == 0:  # base.py:388-392
            G_ = physics_model.NMF_update(W_)
            eval_before = loss(W_, H_)[0]
        else:
            eval_before = eval_after
    elapsed = time.perf_counter() - t0
    if not simplex_H and not simplex_W:
        W_, H_ = rescaled_DH(W_, H_)  # base.py:399-400
    recon, _ = loss(W_, H_)
    if normalize:
        W_ = W_ / norm
    out = dict(W=W_, H=H_, G=G_, GW=G_ @ W_, losses=np.array(losses), detailed_losses=np.array(detailed, dtype=float),
               rel=np.array(rel), n_iter=n_iter, exit=reason, reconstruction_err=recon, snapshots=snaps,
               eval_init=eval_init, const_KL=c_kl, norm_factor=norm)
    out["gammas"] = np.array(gammas, dtype=float)
    if track:
        out.update(angles=np.array(angles, dtype=float), mse=np.array(mses, dtype=float), true_losses=np.array(true_losses, dtype=float))
    if time_iterations:
        out["seconds"] = elapsed
    return out
