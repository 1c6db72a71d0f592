"""Simplex Lagrange-multiplier root finders (espm/estimators/dicotomy.py:4-108) on the GPU."""
import ctypes as C

import numpy as np

from espm_amd.conf import dicotomy_tol, log_shift, maxit_dichotomy


def dichotomy_simplex(num, denum, log_shift=log_shift, tol=dicotomy_tol, maxit=maxit_dichotomy):
    """Solve sum_i max(num_ij / (nu_j + denum_ij), log_shift) = 1 for every column j.

    Same arguments, checks and exceptions as the reference (dicotomy.py:4-55).  The root is found
    per column with a bracketed Newton iteration inside the reference's bracket (the reference
    bisects all columns until the worst one meets ``tol``), so |f(nu)| <= tol holds per column.
    """
    import torch

    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream, require_gpu

    num = np.asarray(num, dtype=np.float64)
    denum = np.asarray(denum, dtype=np.float64)
    assert (num >= 0).all()           # dicotomy.py:17-19
    assert (denum >= 0).all()
    assert (np.sum(num, axis=0) > 0).all()
    if log_shift > 0 and denum.shape[0] * log_shift >= 1:
        raise ValueError("No solution exists!")  # dicotomy.py:22-23
    k, p = num.shape
    den_cols = denum.shape[1]
    if den_cols not in (1, p) or denum.shape[0] != k:
        raise ValueError("denum must be (k, p) or (k, 1)")
    dev = require_gpu()
    d_num = torch.from_numpy(np.ascontiguousarray(num)).to(dev)
    d_den = torch.from_numpy(np.ascontiguousarray(denum)).to(dev)
    d_nu = torch.empty(p, dtype=torch.float64, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.espm_dichotomy_simplex(_ptr(d_num), _ptr(d_den), k, p, den_cols, float(log_shift), float(tol),
                                               int(maxit), _ptr(d_nu), _ptr(d_status), _stream()))
    nu = d_nu.cpu().numpy()
    assert int(d_status.item()) == 0, "dichotomy_simplex preconditions violated on device"
    return nu


def dichotomy_simplex_acc(a, b, minus_c, log_shift=log_shift, tol=dicotomy_tol, maxit=maxit_dichotomy):
    """Multiplier of the quadratic-surrogate H update (dicotomy.py:57-82): for every column j the nu with
    sum_k max(sqrt((b_kj + nu)^2 + 4 a c_kj) - nu - b_kj, 2 a log_shift) = 2 a.  Same arguments and checks as the
    reference; per-column convergence to ``tol`` inside the reference's bracket."""
    import torch

    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream, require_gpu

    b = np.asarray(b, dtype=np.float64)
    minus_c = np.asarray(minus_c, dtype=np.float64)
    assert a >= 0                        # dicotomy.py:66-67
    assert (minus_c >= 0).all()
    if log_shift > 0 and b.shape[0] * log_shift >= 1:
        raise ValueError("No solution exists!")
    k, p = minus_c.shape
    if b.ndim != 2 or b.shape[0] != k or b.shape[1] not in (1, p):
        raise ValueError("b must be (k, p) or (k, 1)")
    dev = require_gpu()
    d_b = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
    d_c = torch.from_numpy(np.ascontiguousarray(minus_c)).to(dev)
    d_nu = torch.empty(p, dtype=torch.float64, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.espm_dichotomy_simplex_acc(float(a), _ptr(d_b), _ptr(d_c), k, p, b.shape[1], float(log_shift), float(tol),
                                                   int(maxit), _ptr(d_nu), _ptr(d_status), _stream()))
    nu = d_nu.cpu().numpy()
    assert int(d_status.item()) == 0, "dichotomy_simplex_acc preconditions violated on device"
    return nu


def dichotomy_simplex_projected_gradient(a, log_shift=log_shift, tol=dicotomy_tol, maxit=maxit_dichotomy):
    """Projection multiplier of the projected-gradient H step (dicotomy.py:84-108): sum_k max(a_kj + nu_j, log_shift) = 1."""
    import torch

    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream, require_gpu

    a = np.asarray(a, dtype=np.float64)
    if log_shift > 0 and a.shape[0] * log_shift >= 1:
        raise ValueError("No solution exists!")
    k, p = a.shape
    dev = require_gpu()
    d_a = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_nu = torch.empty(p, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib.espm_dichotomy_simplex_pg(_ptr(d_a), k, p, float(log_shift), float(tol), int(maxit), _ptr(d_nu), _stream()))
    return d_nu.cpu().numpy()


def dicotomy(a, b, func, maxit, tol):
    """Bisection for func(x) = 0 on arrays of brackets, func(a) > 0 > func(b) (espm/estimators/dicotomy.py:111-173).

    `func` is a Python callable, so this one runs where the arrays are - numpy on the host; the three simplex multipliers above, which
    the reference builds on it, have kernels of their own (the per-pixel root of the H update: csrc/mu_common.hpp::simplex_root).
    The reference's contract, kept: the sign checks of the bracket as assertions; ONE stop rule for all entries (every entry is halved until
    the largest |func| is within tol, or maxit sweeps); the half that keeps the sign change is chosen from func(a) * func(mid) <= 0; `a` and `b`
    are updated in place."""
    f_lo, f_hi = func(a), func(b)
    assert np.sum(f_hi >= 0) == 0
    assert np.sum(f_lo <= 0) == 0
    assert np.sum(np.isnan(f_lo)) == 0
    assert np.sum(np.isnan(f_hi)) == 0
    mid = (a + b) / 2
    f_mid = func(mid)
    sweeps = 0
    while np.max(np.abs(f_mid)) > tol:
        sweeps += 1
        root_left = func(a) * f_mid <= 0          # the sign change lies between a and mid: mid becomes the right end
        np.copyto(b, mid, where=root_left)
        np.copyto(a, mid, where=np.logical_not(root_left))
        mid = (a + b) / 2
        f_mid = func(mid)
        if sweeps >= maxit:
            print("Dicotomy stopped for maximum number of iterations with an error of : {}".format(np.max(np.abs(f_mid))))
            break
    return mid
