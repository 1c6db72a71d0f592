"""Sparse-aware fp64 variant of the oracle, for parity checks at BASELINE.json's FULL sizes (TEST INFRASTRUCTURE, see
``oracle/__init__.py``).

``oracle.mu_oracle`` restates the reference op for op - dense ``Y = GW @ H`` and ``X / Y`` of n x p doubles (4.3 GB each
at 2048 x 512^2) and a dense identity G - and therefore cannot run at the headline size in a test.  This module
evaluates the SAME update rules (espm/estimators/updates.py:6-78, :83-156, the loss of base.py:167-207 and
smooth_nmf.py:457-475) touching only the non-zero entries of X:

* ``Y`` is formed only where ``X != 0`` (``R = X / Y`` is zero elsewhere, and so is ``X log Y``);
* ``sum(Y) = colsum(GW) . rowsum(H)`` (exact algebra);
* the W numerator is associated ``G^T (R H^T)`` instead of ``(G^T R) H^T`` (identical up to rounding, updates.py:58-59);
* the ``eps * log Y`` contribution of the empty bins to the KL term (measures.py:493-504 with ``max(X, eps)``) is dropped:
  <= 1e-12 relative at any size used here, and it is reported by ``dropped_eps_logy`` on request.

Everything else - the global-stop bisection, the un-clamped ``H @ L``, the clamp at ``log_shift``, ``fixed_*`` - is
the faithful oracle's own code (imported from ``oracle.mu_oracle``).  PINNED: ``tests/test_oracle_golden.py::
test_sparse_oracle_on_f6`` requires the same trajectories as the faithful oracle (and hence as the reference-generated
fixture F6) to 1e-9.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import mu_oracle as oc


def _workers():
    return max(1, min(16, os.cpu_count() or 1))


class SparseX:
    """The non-zero entries of X (n, p): ``ch``, ``px`` (int32), ``val`` (float64), sorted by pixel, cut into chunks at
    pixel boundaries.  The three passes over the entries are mapped over the chunks by a thread pool (numpy releases the
    GIL in take / multiply / bincount): at 2048 x 512^2 with 21 % non-zero entries an iteration takes seconds, not a minute."""

    def __init__(self, ch, px, val, n, p, chunk=1 << 22):
        px = np.asarray(px)
        if px.size > 1 and not bool(np.all(px[1:] >= px[:-1])):
            order = np.argsort(px, kind="stable")
            ch, px, val = np.asarray(ch)[order], px[order], np.asarray(val)[order]
        self.ch = np.ascontiguousarray(ch, dtype=np.int32)
        self.px = np.ascontiguousarray(px, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.n, self.p = int(n), int(p)
        self.sum_x = float(self.val.sum())
        self.const_kl = float(np.sum(self.val * np.log(self.val)) - self.sum_x)   # base.py:200-201 (0 log 0 = 0)
        # chunks [lo, hi) of entries that end at pixel boundaries, with their pixel ranges [q0, q1)
        nnz = self.val.shape[0]
        cuts = [0]
        while cuts[-1] < nnz:
            hi = min(nnz, cuts[-1] + chunk)
            if hi < nnz:
                hi = int(np.searchsorted(self.px, self.px[hi], side="left"))   # back to the start of that pixel
                if hi <= cuts[-1]:
                    hi = int(np.searchsorted(self.px, self.px[cuts[-1]], side="right"))
            cuts.append(hi)
        self.chunks = [(lo, hi, int(self.px[lo]), int(self.px[hi - 1]) + 1) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]

    @classmethod
    def from_dense(cls, X):
        X = np.asarray(X)
        px, ch = np.nonzero(X.T)          # pixel-major walk: sorted by pixel
        return cls(ch, px, X[ch, px], X.shape[0], X.shape[1])

    def _map(self, fn):
        if len(self.chunks) == 1 or _workers() == 1:
            return [fn(c) for c in self.chunks]
        with ThreadPoolExecutor(_workers()) as ex:
            return list(ex.map(fn, self.chunks))

    def _rows(self, GW, HT, lo, hi):
        """The GW rows and the H columns of the entries [lo, hi) as (m, k) arrays, and (GW @ H) there."""
        Gc = np.take(GW, self.ch[lo:hi], axis=0)
        Hq = np.take(HT, self.px[lo:hi], axis=0)
        return Gc, Hq, np.einsum("ik,ik->i", Gc, Hq)

    def _starts(self, lo, hi, q0, q1):
        """First entry (relative to lo) of every pixel q0..q1 (the entries are sorted by pixel)."""
        return np.searchsorted(self.px[lo:hi], np.arange(q0, q1 + 1, dtype=np.int32), side="left")

    def gwt_r(self, GW, H):
        """GW^T (X / (GW H)) (k, p)."""
        GW, HT = np.ascontiguousarray(GW), np.ascontiguousarray(H.T)
        out = np.zeros((self.p, GW.shape[1]))

        def work(chunk):
            lo, hi, q0, q1 = chunk
            Gc, _, y = self._rows(GW, HT, lo, hi)
            Gc *= (self.val[lo:hi] / y)[:, None]
            st = self._starts(lo, hi, q0, q1)
            seg = np.add.reduceat(Gc, np.minimum(st[:-1], hi - lo - 1), axis=0)
            seg[st[1:] == st[:-1]] = 0.0                     # pixels without an entry (reduceat returns a row there)
            out[q0:q1] = seg
        self._map(work)
        return np.ascontiguousarray(out.T)

    def r_ht(self, GW, H):
        """(X / (GW H)) H^T (n, k)."""
        GW, HT = np.ascontiguousarray(GW), np.ascontiguousarray(H.T)

        def work(chunk):
            lo, hi, _, _ = chunk
            _, Hq, y = self._rows(GW, HT, lo, hi)
            Hq *= (self.val[lo:hi] / y)[:, None]
            c = self.ch[lo:hi]
            part = np.empty((self.n, HT.shape[1]))
            for kk in range(HT.shape[1]):
                part[:, kk] = np.bincount(c, weights=Hq[:, kk], minlength=self.n)
            return part
        return np.sum(self._map(work), axis=0)

    def x_log_y(self, GW, H):
        """sum X log(GW H) over the stored entries."""
        GW, HT = np.ascontiguousarray(GW), np.ascontiguousarray(H.T)

        def work(chunk):
            lo, hi, _, _ = chunk
            return float(np.dot(self.val[lo:hi], np.log(self._rows(GW, HT, lo, hi)[2])))
        return float(np.sum(self._map(work)))


def step_h(sx, G, W, H, simplex_H=False, mu=0, log_shift=oc.LOG_SHIFT, epsilon_reg=1, dicotomy_tol=oc.DICOTOMY_TOL,
           lambda_L=0, L=None, sigmaL=oc.SIGMA_L, fixed_H=None, exact_root=False):
    """multiplicative_step_h, KL branch (updates.py:83-156), on the stored entries; G may be None (identity)."""
    if lambda_L != 0:
        if L is None:
            raise ValueError("Please provide the laplacian")
        HL = H @ L
    GW = W if G is None else G @ W
    num = sx.gwt_r(GW, H)
    den = GW.sum(axis=0)[:, None]
    if not (np.isscalar(mu) and mu == 0):
        mu_col = np.asarray(mu, dtype=float)
        if mu_col.ndim == 1:
            mu_col = mu_col[:, None]
        den = den + mu_col / (H + epsilon_reg)
    if lambda_L != 0:
        maxH = H.max(axis=1, keepdims=True)
        num = num + lambda_L * sigmaL * maxH
        den = den + lambda_L * sigmaL * maxH + lambda_L * HL
    num = H * num
    if simplex_H and exact_root:
        delta, e = oc.dichotomy_simplex_exact(num, den, log_shift)
        with np.errstate(divide="ignore", invalid="ignore"):
            new_H = np.fmax(num / (delta + e), log_shift)
    else:
        nu = oc.dichotomy_simplex(num, den, log_shift=log_shift, tol=dicotomy_tol) if simplex_H else 0
        new_H = np.maximum(num / (den + nu), log_shift)
    if fixed_H is not None:
        keep = fixed_H >= 0
        new_H[keep] = fixed_H[keep]
    return new_H


def step_w(sx, G, W, H, simplex_W=False, log_shift=oc.LOG_SHIFT, fixed_W=None, simplex_rows=None):
    """multiplicative_step_w, KL branch (updates.py:6-78)."""
    GW = W if G is None else G @ W
    A = sx.r_ht(GW, H)                               # R H^T (n, k)
    gta = A if G is None else G.T @ A                # G^T (R H^T) == (G^T R) H^T
    num = W * gta
    colsum_g = np.ones(W.shape[0]) if G is None else G.sum(axis=0)
    den = colsum_g[:, None] @ H.sum(axis=1)[None, :]
    if simplex_W:
        if simplex_rows is not None:
            nu = oc.dichotomy_simplex(num[simplex_rows, :], den[simplex_rows, :], log_shift=log_shift, tol=oc.DICOTOMY_TOL)
            den[simplex_rows, :] = den[simplex_rows, :] + nu
        else:
            den = den + oc.dichotomy_simplex(num, den, log_shift=log_shift, tol=oc.DICOTOMY_TOL)
    new_W = np.maximum(num / den, log_shift)
    if fixed_W is not None:
        keep = fixed_W >= 0
        new_W[keep] = fixed_W[keep]
    return new_W


def loss(sx, G, W, H, L, mu=0, epsilon_reg=1, lambda_L=0.0, log_shift=oc.LOG_SHIFT, average=True):
    """SmoothNMF.loss (smooth_nmf.py:457-475, base.py:196-203): (total, [lkl, reg, lap])."""
    GW = np.maximum(W if G is None else G @ W, log_shift)
    Hc = np.maximum(H, log_shift)
    sum_y = float(GW.sum(axis=0) @ Hc.sum(axis=1))
    lkl = sum_y - sx.x_log_y(GW, Hc) + sx.const_kl
    reg = oc.log_reg(H, mu, epsilon_reg)
    lap = 0.5 * lambda_L * oc.trace_xtLx(L, H.T)
    numel = float(sx.n) * float(sx.p) if average else 1.0
    return (lkl + reg + lap) / numel, [lkl / numel, reg / numel, lap / numel]


def fit(sx, n_components, G=None, W=None, H=None, *, lambda_L=0.0, mu=0, epsilon_reg=1, simplex_H=False, simplex_W=True,
        shape_2d=None, tol=1e-4, max_iter=200, log_shift=oc.LOG_SHIFT, dicotomy_tol=oc.DICOTOMY_TOL, gamma=None,
        fixed_H=None, fixed_W=None, exact_root=False, record_at=()):
    """The fit loop of ``oracle.mu_oracle.fit`` with ``no_stop_criterion=True`` (base.py:313-394), default solver, W and
    H given (the callers hand over the initial state: the data must have no all-zero channel or pixel, which the
    reference would fill with log_shift, base.py:519-528)."""
    if simplex_H and simplex_W:
        simplex_W, simplex_H = True, False
    W_, H_ = np.array(W, dtype=np.float64), np.array(H, dtype=np.float64)
    L_ = oc.laplacian_matrix(*shape_2d) if shape_2d is not None else oc.identity_L(sx.p)
    gamma_ = oc.SIGMA_L if gamma is None else gamma
    losses, detailed, rel, snaps = [], [], [], {}
    eval_init, _ = loss(sx, G, W_, H_, L_, mu, epsilon_reg, lambda_L, log_shift)
    for it in range(1, max_iter + 1):
        old_W, old_H = W_, H_
        H_ = step_h(sx, G, W_, H_, simplex_H=simplex_H, mu=mu, log_shift=log_shift, epsilon_reg=epsilon_reg,
                    dicotomy_tol=dicotomy_tol, lambda_L=lambda_L, L=L_, sigmaL=gamma_, fixed_H=fixed_H, exact_root=exact_root)
        W_ = step_w(sx, G, W_, H_, simplex_W=simplex_W, log_shift=log_shift, fixed_W=fixed_W)
        ev, det = loss(sx, G, W_, H_, L_, mu, epsilon_reg, lambda_L, log_shift)
        losses.append(ev)
        detailed.append(det)
        rel.append([np.max(np.abs(W_ - old_W) / (W_ + tol * np.mean(W_))), np.max(np.abs(H_ - old_H) / (H_ + tol * np.mean(H_)))])
        if it in record_at:
            snaps[it] = (W_.copy(), H_.copy())
    return dict(W=W_, H=H_, GW=W_ if G is None else G @ W_, losses=np.array(losses), detailed_losses=np.array(detailed, dtype=float),
                rel=np.array(rel), n_iter=max_iter, eval_init=eval_init, snapshots=snaps)


def dropped_eps_logy(sx, G, W, H, log_shift=oc.LOG_SHIFT):
    """Upper bound of what the empty bins would add to the KL term: eps * sum over ALL (c, j) of |log Y_cj| <=
    eps * n p max|log Y|, with Y bounded through its factors."""
    GW = np.maximum(W if G is None else G @ W, log_shift)
    Hc = np.maximum(H, log_shift)
    hi = np.log(max(GW.max() * Hc.max() * H.shape[0], 1.0 + 1e-300))
    lo = abs(np.log(GW.min() * Hc.min()))
    return log_shift * float(sx.n) * float(sx.p) * max(hi, lo)
