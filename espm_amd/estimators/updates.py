"""Multiplicative update rules with the reference's module-level signatures
(espm/estimators/updates.py:6-16, :83, :160), running on the GPU."""
import numpy as np

from espm_amd.conf import dicotomy_tol, log_shift, sigmaL
from espm_amd.utils import classify_laplacian


def _as_identity(G):
    G = np.asarray(G)
    return G.ndim == 2 and G.shape[0] == G.shape[1] and np.array_equal(G, np.eye(G.shape[0], dtype=G.dtype))


def _safe_inputs(G, W, H, log_shift, safe):
    if safe:  # updates.py:20-27, :98-105
        assert np.sum(H < -log_shift / 2) == 0
        assert np.sum(W < -log_shift / 2) == 0
        assert np.sum(np.asarray(G) < -log_shift / 2) == 0
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    return W, H


def _engine(X, G, W, H, **kw):
    from espm_amd.engine import MUEngine

    X = np.asarray(X)
    ident = _as_identity(G)
    eng = MUEngine(X, H.shape[0], G=None if ident else np.asarray(G), fix_zero_lines=False, max_iter=1, **kw)
    eng.load_state(W, H)
    return eng


def multiplicative_step_w(X, G, W, H, simplex_W=False, log_shift=log_shift, safe=True, l2=False, fixed_W=None,
                          physics_model=None, use_bregman=False):
    """Multiplicative step in W (espm/estimators/updates.py:6-78: KL branch, and the Frobenius branch l2=True :31-36)."""
    W = np.asarray(W)
    H = np.asarray(H)
    W, H = _safe_inputs(G, W, H, log_shift, safe)
    if use_bregman and not l2:  # updates.py:40-48: no simplex in this branch either; G = identity only (engine)
        eng = _engine(X, G, W, H, simplex_H=False, simplex_W=False, log_shift=log_shift, fixed_W=fixed_W, bregman=True)
        out = eng.step_w_only()
        return out.astype(np.result_type(W.dtype, np.float32) if W.dtype == np.float32 else np.float64)
    if l2:  # new_W = W / (G^T G W H H^T) * (G^T X H^T): no simplex in this branch
        eng = _engine(X, G, W, H, simplex_H=False, simplex_W=False, log_shift=log_shift, fixed_W=fixed_W, x_store="f32")
        out = eng.step_w_only(l2=True)
        return out.astype(np.result_type(W.dtype, np.float32) if W.dtype == np.float32 else np.float64)
    rows = physics_model.NMF_simplex() if (simplex_W and physics_model is not None) else None
    eng = _engine(X, G, W, H, simplex_H=False, simplex_W=simplex_W, log_shift=log_shift, fixed_W=fixed_W,
                  simplex_rows=rows)
    out = eng.step_w_only()
    return out.astype(np.result_type(W.dtype, np.float32) if W.dtype == np.float32 else np.float64)


def multiplicative_step_wq(X, G, W, H, simplex_W=True, log_shift=log_shift, safe=True, physics_model=None):
    """Multiplicative step in W "using the WQ technique" (espm/estimators/updates.py:232-261), as the reference runs it.

    By its docstring exactly multiplicative_step_w, and without the simplex it is (term1 = G^T (XQ / (GW + log_shift)) is G^T ((X / GWH) H^T)
    with log_shift added instead of clamped).  WITH simplex_W - its default - the reference finds the multiplier from
    dichotomy_simplex(term1, term2), the numerators without their factor W (updates.py:253-258), so W' = W / (term2 + nu) * term1 is not on the
    simplex (column sums 0.5-0.9 on the fixture F19); that is what comes back here too.  The streaming part - term1 - is the HIP W step without a
    simplex (W'_free = W term1 / term2), the multiplier the module's dichotomy_simplex; nothing in a fit calls this function."""
    W = np.asarray(W)
    H = np.asarray(H)
    if safe:  # asserts only, no clamping (updates.py:239-243)
        assert np.sum(H < -log_shift / 2) == 0
        assert np.sum(W < -log_shift / 2) == 0
        assert np.sum(np.asarray(G) < -log_shift / 2) == 0
    free = multiplicative_step_w(X, G, W, H, simplex_W=False, log_shift=log_shift, safe=False)
    if not simplex_W:
        return free
    from espm_amd.estimators.dicotomy import dichotomy_simplex
    Gd = np.asarray(G() if callable(G) else G, dtype=np.float64)
    term2 = np.sum(Gd, axis=0, keepdims=True).T @ np.sum(H.astype(np.float64), axis=1, keepdims=True).T   # updates.py:251
    term1 = free.astype(np.float64) * term2 / np.maximum(W.astype(np.float64), log_shift)
    if physics_model is not None:
        rows = physics_model.NMF_simplex()
        term2[rows, :] = term2[rows, :] + dichotomy_simplex(term1[rows, :], term2[rows, :], log_shift=log_shift, tol=dicotomy_tol)
    else:
        term2 = term2 + dichotomy_simplex(term1, term2, log_shift=log_shift, tol=dicotomy_tol)
    return (W / term2 * term1).astype(free.dtype)


def multiplicative_step_h(X, G, W, H, simplex_H=False, mu=0, log_shift=log_shift, epsilon_reg=1, safe=True,
                          dicotomy_tol=dicotomy_tol, lambda_L=0, L=None, l2=False, sigmaL=sigmaL, fixed_H=None,
                          use_bregman=False):
    """Multiplicative step in H (espm/estimators/updates.py:83-156: KL branch, and the Frobenius branch l2=True :109-118)."""
    shape_2d = None
    W = np.asarray(W)
    H = np.asarray(H)
    if not (lambda_L == 0):
        if L is None:
            raise ValueError("Please provide the laplacian")  # updates.py:94-95
        kind, shape_2d = classify_laplacian(L, H.shape[1])
    W, H = _safe_inputs(G, W, H, log_shift, safe)
    if l2:  # updates.py:109-118
        assert lambda_L == 0
        assert np.all(np.asarray(mu) == 0)
        eng = _engine(X, G, W, H, simplex_H=simplex_H, simplex_W=False, log_shift=log_shift, dicotomy_tol=dicotomy_tol,
                      fixed_H=fixed_H, compute_loss=False, x_store="f32")
        out = eng.step_h_only(l2=True)
        if eng.bad_count() > 0 and safe:
            raise AssertionError("multiplicative_step_h: non-finite update or simplex preconditions violated")
        return out.astype(np.float32 if H.dtype == np.float32 else np.float64)
    eng = _engine(X, G, W, H, simplex_H=simplex_H, simplex_W=False, mu=mu, log_shift=log_shift,
                  epsilon_reg=epsilon_reg, dicotomy_tol=dicotomy_tol, lambda_L=lambda_L, shape_2d=shape_2d,
                  sigmaL=sigmaL, fixed_H=fixed_H, compute_loss=False, bregman=bool(use_bregman))   # (updates.py:120-125)
    out = eng.step_h_only()
    if eng.bad_count() > 0 and safe:
        raise AssertionError("multiplicative_step_h: non-finite update or simplex preconditions violated")
    return out.astype(np.float32 if H.dtype == np.float32 else np.float64)


def multiplicative_step_hq(X, G, W, H, simplex_H=True, log_shift=log_shift, safe=True, dicotomy_tol=dicotomy_tol, lambda_L=0,
                           L=None, sigmaL=sigmaL, fixed_H=None):
    """Multiplicative step in H from the quadratic surrogate of the Laplacian term (espm/estimators/updates.py:263-315)."""
    shape_2d = None
    W = np.asarray(W)
    H = np.asarray(H)
    if not (lambda_L == 0):
        if L is None:
            raise ValueError("Please provide the laplacian")
        kind, shape_2d = classify_laplacian(L, H.shape[1])
    if safe:  # updates.py:279-283 (asserts only: no clamping in this function)
        assert np.sum(H < -log_shift / 2) == 0
        assert np.sum(W < -log_shift / 2) == 0
        assert np.sum(np.asarray(G) < -log_shift / 2) == 0
    eng = _engine(X, G, W, H, simplex_H=simplex_H, simplex_W=False, log_shift=log_shift, dicotomy_tol=dicotomy_tol,
                  lambda_L=lambda_L, shape_2d=shape_2d, sigmaL=sigmaL, fixed_H=fixed_H, compute_loss=False, h_rule=1)
    out = eng.step_h_only()
    if eng.bad_count() > 0 and safe:
        raise AssertionError("multiplicative_step_hq: non-finite update or preconditions violated")
    return out.astype(np.float32 if H.dtype == np.float32 else np.float64)


def estimate_Lipschitz_bound_w(log_shift, X, G, k):
    """Lipschitz bound of the KL gradient in W at the corner W = H = log_shift (espm/estimators/updates.py:397-407): the
    default gamma_W of the projected gradient (of the order of X / log_shift^3: the iterates do not move with it).  One
    elementwise pass over X on the host, once per fit."""
    if G is None:
        G = np.eye(X.shape[0])
    Wlim = np.ones([G.shape[1], k]) * log_shift
    Hlim = np.ones([k, X.shape[1]]) * log_shift
    DH = (G @ Wlim) @ Hlim
    return np.max((np.sum(Hlim, axis=0, keepdims=True) * X / (DH ** 2)) @ Hlim.T)


def estimate_Lipschitz_bound_h(log_shift, X, G, k, lambda_L=0, mu=0, epsilon_reg=1):
    """Lipschitz bound of the gradient in H at the corner W = H = log_shift (espm/estimators/updates.py:409-419)."""
    if G is None:
        G = np.eye(X.shape[0])
    Wlim = np.ones([G.shape[1], k]) * log_shift
    Hlim = np.ones([k, X.shape[1]]) * log_shift
    D = G @ Wlim
    DH = D @ Hlim
    return np.max(D.T @ (np.sum(D, axis=1, keepdims=True) * X / (DH ** 2))) + 2 * lambda_L + mu * epsilon_reg


def _dev64(*arrays):
    import torch

    from espm_amd.engine import require_gpu
    dev = require_gpu()
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev) for a in arrays]


def update_q(D, H, log_shift=log_shift):
    """The "Q step" (espm/estimators/updates.py:225-230): Q[i, j, k] = H[k, j] D[i, k] / ((D H)[i, j] + log_shift), an
    (n, p, k) array formed on the device."""
    Dd, Hd = _dev64(D, H)
    q = Hd.t().unsqueeze(0) * (Dd.unsqueeze(1) / ((Dd @ Hd).unsqueeze(2) + log_shift))
    return q.cpu().numpy()


def gradW(X, G, W, H, log_shift=log_shift, safe=False, l2=False):
    """Gradient of the data term in W (espm/estimators/updates.py:303-313): G^T (-(X / GWH) H^T + rowsum(H)^T), or
    2 G^T (GWH - X) H^T with l2.  fp64 on the device."""
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    Xd, Gd, Wd, Hd = _dev64(X, G, W, H)
    if l2:
        grad = 2 * Gd.t() @ (((Gd @ Wd) @ Hd - Xd) @ Hd.t())
    else:
        grad = Gd.t() @ (-(Xd / ((Gd @ Wd) @ Hd)) @ Hd.t() + Hd.sum(dim=1, keepdim=True).t())
    return grad.cpu().numpy()


def gradH(X, G, W, H, mu=0, lambda_L=0, L=None, epsilon_reg=1, log_shift=log_shift, safe=False, l2=False):
    """Gradient of the regularised loss in H (espm/estimators/updates.py:315-342): -(GW)^T (X / GWH) + colsum(GW)^T (or
    (GW)^T (GWH - X) with l2) + mu / (H + epsilon_reg) + lambda_L (L H^T)^T.  fp64 on the device; L, whatever matrix the
    caller hands over, is applied on the host."""
    if not (lambda_L == 0) and L is None:
        raise ValueError("Please provide the laplacian")
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    Xd, Gd, Wd, Hd = _dev64(X, G, W, H)
    D = Gd @ Wd
    if l2:
        grad = D.t() @ (D @ Hd - Xd)
    else:
        grad = -D.t() @ (Xd / (D @ Hd)) + D.sum(dim=0, keepdim=True).t()
    grad = grad.cpu().numpy()
    if not (np.isscalar(mu) and mu == 0):
        if len(np.shape(mu)) == 1:
            mu = np.expand_dims(mu, axis=1)
        grad += mu / (np.asarray(H) + epsilon_reg)
    if not (lambda_L == 0):
        grad += (lambda_L * L @ np.asarray(H).T).T
    return grad


def proj_grad_step_h(X, G, W, H, gamma, simplex_H=True, mu=0, log_shift=log_shift, epsilon_reg=1, safe=True,
                     dicotomy_tol=dicotomy_tol, lambda_L=0, L=None, l2=False, fixed_H=None):
    """Projected-gradient step in H (espm/estimators/updates.py:372-395).  The KL branch is the engine's h_rule = 2; with l2=True - a
    branch only a direct call reaches, smooth_nmf.py:233-237 - the step is composed of the module-level pieces, as the reference
    composes it: gradH(l2=True), the gradient step, dichotomy_simplex_projected_gradient, the clamp, fixed_H."""
    if l2:
        if safe:  # updates.py:376-378
            H = np.maximum(H, log_shift)
            W = np.maximum(W, log_shift)
        from espm_amd.estimators.dicotomy import dichotomy_simplex_projected_gradient
        new_H = np.asarray(H) - 1 / gamma * gradH(X, G, W, H, mu=mu, lambda_L=lambda_L, L=L, epsilon_reg=epsilon_reg, log_shift=log_shift, safe=safe, l2=True)
        nu = dichotomy_simplex_projected_gradient(new_H, log_shift=log_shift, tol=dicotomy_tol) if simplex_H else 0
        new_H = np.maximum(new_H + nu, log_shift)
        if fixed_H is not None:
            new_H[fixed_H >= 0] = fixed_H[fixed_H >= 0]
        return new_H
    shape_2d = None
    W = np.asarray(W)
    H = np.asarray(H)
    if not (lambda_L == 0):
        if L is None:
            raise ValueError("Please provide the laplacian")
        kind, shape_2d = classify_laplacian(L, H.shape[1])
    if safe:  # updates.py:376-378
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    eng = _engine(X, G, W, H, simplex_H=simplex_H, simplex_W=False, mu=mu, log_shift=log_shift, epsilon_reg=epsilon_reg,
                  dicotomy_tol=dicotomy_tol, lambda_L=lambda_L, shape_2d=shape_2d, sigmaL=float(gamma), fixed_H=fixed_H,
                  compute_loss=False, h_rule=2)
    out = eng.step_h_only()
    return out.astype(np.float32 if H.dtype == np.float32 else np.float64)


def proj_grad_step_w(X, G, W, H, gamma, simplex_W=True, log_shift=log_shift, safe=True, l2=False, fixed_W=None):
    """Projected-gradient step in W (espm/estimators/updates.py:353-370; l2=True: composed of gradW(l2=True), the step, the clamp and
    fixed_W, as the reference composes it)."""
    if l2 and not simplex_W:
        if safe:
            H = np.maximum(H, log_shift)
            W = np.maximum(W, log_shift)
        new_W = np.maximum(np.asarray(W) - 1 / gamma * gradW(X, G, W, H, log_shift=log_shift, safe=safe, l2=True), log_shift)
        if fixed_W is not None:
            new_W[fixed_W >= 0] = fixed_W[fixed_W >= 0]
        return new_W
    if simplex_W:  # updates.py:368-369 (the reference raises after the step; here before)
        raise NotImplementedError("Simplex constraint not implemented for W using the projected gradient method")
    W = np.asarray(W)
    H = np.asarray(H)
    if safe:
        H = np.maximum(H, log_shift)
        W = np.maximum(W, log_shift)
    eng = _engine(X, G, W, H, simplex_H=False, simplex_W=False, log_shift=log_shift, fixed_W=fixed_W, pg_gamma_w=float(gamma))
    out = eng.step_w_only()
    return out.astype(np.result_type(W.dtype, np.float32) if W.dtype == np.float32 else np.float64)


def _initial_factors(X, n_components, init, random_state, sklearn_init, X_device=None, X_mean=None, shard=None):
    from espm_amd import init_device

    nndsvd = init in (None, "nndsvd", "nndsvda", "nndsvdar") and n_components <= min(X.shape)
    if nndsvd and X.size >= init_device.DEVICE_INIT_MIN_SIZE:
        import torch
        if torch.cuda.is_available():
            if shard is not None and X.shape[0] >= X.shape[1]:   # (more channels than pixels: the sharded routine does not apply, and the image is small)
                shard, X_device = None, None
            return init_device.initialize_nmf_device(X, n_components, init=init, random_state=random_state,
                                                     X_device=X_device, X_mean=X_mean, shard=shard)
    return sklearn_init(X, n_components=n_components, init=init, random_state=random_state)


def initialize_algorithms(X, G, W, H, n_components, init, random_state, simplex_H, simplex_W, logshift=log_shift,
                          physics_model=None, X_device=None, X_mean=None, shard=None):
    """Initial G, W, H (espm/estimators/updates.py:160-223).

    Like the reference: scikit-learn's NNDSVD / random initialisation and small least-squares fits, once per
    fit, outside the multiplicative-update loop.  For a large X the NNDSVD's sixteen passes over X run on the
    GPU (espm_amd/init_device.py: same algorithm, same random stream, result equal to scikit-learn's to rounding)."""
    from sklearn.decomposition._nmf import _initialize_nmf

    if G is None:
        skip_second = True
        G = np.diag(np.ones(X.shape[0]).astype(X.dtype))  # updates.py:163-166 (dense identity)
    else:
        skip_second = False
    if W is None:
        if H is None:
            # (shard: X_device is one rank's block of pixels of a sharded fit, espm_amd/estimators/base.py - the factors come back whole)
            D, H = _initial_factors(X, n_components, init, random_state, _initialize_nmf, X_device, X_mean, shard)
            if simplex_H:
                H = np.nan_to_num(H, nan=1.0 / H.shape[0])
                scale = np.sum(H, axis=0, keepdims=True)
                H = H / scale
                D = D * np.mean(scale)
        else:
            D = np.abs(np.linalg.lstsq(H.T, X.T, rcond=None)[0].T)
        if skip_second:
            W = D
        elif physics_model is not None:
            W = physics_model.NMF_initialize_W(D)
            if simplex_W:
                indices = physics_model.NMF_simplex()
                W = np.nan_to_num(W, nan=1.0 / W.shape[0])
                W[indices, :] = W[indices, :] / np.sum(W[indices, :], axis=0, keepdims=True)
        else:
            W = np.abs(np.linalg.lstsq(G, D, rcond=None)[0])
            if simplex_W:
                W = np.nan_to_num(W, nan=1.0 / W.shape[0])
                W = W / np.sum(W, axis=0, keepdims=True)
    elif H is None:
        D = G @ W
        H = np.abs(np.linalg.lstsq(D, X, rcond=None)[0])
        if simplex_H:
            H = H / np.sum(H, axis=0, keepdims=True)
    W = np.maximum(W, logshift)
    H = np.maximum(H, logshift)
    return G, W, H
