"""Surrogates of the Laplacian term (espm/estimators/surrogates.py:6-170) with the reference's module-level signatures.

The sums over the (k, p) arrays - tr(Ht L Ht^T), tr(Ht L H^T), tr(H L H^T), ||Ht - H||^2 and the per-component generalised KL
divergences - come from one pass of ``espm_surrogate_terms`` on the GPU (the kernel the linesearch of a fit uses)."""
import numpy as np

from espm_amd.conf import sigmaL
from espm_amd.utils import classify_laplacian


def _terms(Ht, H, L):
    """(t1, t2, b, sq, dg[k]) for Ht, H (k, p) and the Laplacian L (grid or identity)."""
    import torch

    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream, require_gpu

    Ht = np.asarray(Ht)
    H = np.asarray(H)
    k, p = Ht.shape
    V = _lib.variant(k)   # (raises beyond 16 components)
    kind, shape = classify_laplacian(L, p)
    dev = require_gpu()
    a = torch.from_numpy(np.ascontiguousarray(Ht, dtype=np.float32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(H, dtype=np.float32)).to(dev)
    nparts = (4 + V.KP) * ((p + 511) // 512)
    part = torch.zeros(nparts, dtype=torch.float64, device=dev)
    out = torch.zeros(4 + V.KP, dtype=torch.float64, device=dev)
    grid = 0 if kind == "identity" else 1
    nx, ny = (0, 0) if kind == "identity" else shape
    V.check(V.lib.espm_surrogate_terms(_ptr(a), _ptr(b), k, p, p, int(nx), int(ny), grid, _ptr(part), nparts, _ptr(out), _stream()))
    t = out.cpu().numpy()
    return t[0], t[1], t[2], t[3], t[4:4 + k]


def smooth_l2_surrogate(Ht, L, H=None, sigmaL=sigmaL, lambda_L=1):
    """lambda/2 (2 tr(Ht L H^T) - tr(Ht L Ht^T) + sigma ||Ht - H||^2), surrogates.py:6-58."""
    t1, t2, _, sq, _ = _terms(Ht, Ht if H is None else H, L)
    return lambda_L / 2 * ((2 * t2 - t1 + sigmaL * sq) if H is not None else t1)


def smooth_dgkl_surrogate(Ht, L, H=None, sigmaL=sigmaL, lambda_L=1):
    """lambda/2 (2 tr(Ht L H^T) - tr(Ht L Ht^T) + sigma sum_k max_j H_kj sum_j dgkl(Ht_kj, H_kj)), surrogates.py:65-114."""
    if H is None:
        return lambda_L / 2 * _terms(Ht, Ht, L)[0]
    t1, t2, _, _, dg = _terms(Ht, H, L)
    t3 = float(np.sum(np.max(np.asarray(H), axis=1) * dg))
    return lambda_L / 2 * (2 * t2 - t1 + sigmaL * t3)


def diff_surrogate(Ht, H, L, sigmaL=sigmaL, lambda_L=1, algo="log_surrogate"):
    """Surrogate minus the Laplacian term at H, surrogates.py:116-149."""
    t1, t2, b, sq, dg = _terms(Ht, H, L)
    if algo in ("log_surrogate", "bmd"):
        t3 = float(np.sum(np.max(np.asarray(H), axis=1) * dg))
    elif algo == "l2_surrogate":
        t3 = sq
    else:
        raise ValueError("Unknown algorithm")
    return lambda_L / 2 * (2 * t2 - t1 + sigmaL * t3) - b * lambda_L / 2


def quadratic_surrogate(x, xt, f_xt, gradf_xt, sigma):
    """f(xt) + <x - xt, grad f(xt)> + sigma ||x - xt||^2, surrogates.py:153-170 (element-wise sums on the device)."""
    import torch

    from espm_amd.engine import require_gpu
    dev = require_gpu()
    d = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64) - np.asarray(xt, dtype=np.float64))).to(dev)
    g = torch.from_numpy(np.ascontiguousarray(gradf_xt, dtype=np.float64)).to(dev)
    return float(f_xt + (d * g).sum() + sigma * (d * d).sum())
