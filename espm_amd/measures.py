"""Objective pieces of the path (espm/measures.py:456-504, :524-548, :560-577) on the GPU."""
import ctypes as C

import numpy as np

from espm_amd.conf import log_shift


def KLdiv_loss(X, W, H, log_shift=log_shift, average=False):
    """sum(WH) - sum(max(X, eps) log(WH)) with W, H clamped at eps (espm/measures.py:456-504).

    Evaluated by the H-step kernel in loss-only mode: it forms sum X log(X / Y) per pixel tile;
    sum X log X is added back here."""
    import torch

    from espm_amd.engine import MUEngine

    X = np.asarray(X)
    W = np.maximum(np.asarray(W), log_shift)
    H = np.maximum(np.asarray(H), log_shift)
    eng = MUEngine(X, H.shape[0], fix_zero_lines=False, max_iter=1, log_shift=log_shift, simplex_W=False)
    eng.load_state(W, H)
    eng.eval_current(advance_h=False)
    h = eng.history(average=False)
    kl_div = float(h["kl"][0])                      # sum X ln(X/Y) + sum Y - sum X
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64))
    xlogx = float(torch.xlogy(Xd, Xd.clamp_min(log_shift)).sum())
    val = kl_div + eng.sum_x - xlogx                # (eps * log Y on empty bins is below the sum's resolution)
    return val / X.size if average else val


def log_reg(H, mu, epsilon=1, average=False):
    """sum_ij mu_i log(H_ij + eps) (espm/measures.py:524-548); a (k, p) elementwise reduction done with
    torch on the device (plumbing - the fit loop gets this term from the fused H-step)."""
    import torch

    from espm_amd.engine import require_gpu

    dev = require_gpu()
    Hd = torch.from_numpy(np.ascontiguousarray(H, dtype=np.float64)).to(dev)
    mud = torch.as_tensor(np.asarray(mu, dtype=np.float64), device=dev)
    if mud.dim() == 1:
        mud = mud[:, None]
    t = mud * torch.log(Hd + epsilon)
    return float(t.mean() if average else t.sum())


def trace_xtLx(L, x, average=False):
    """Tr(x^T L x) = sum(x * (L x)) (espm/measures.py:560-577) with the device stencil."""
    import torch

    from espm_amd import _lib
    from espm_amd.engine import _ptr, _stream, require_gpu
    from espm_amd.utils import classify_laplacian

    x = np.asarray(x)
    xm = x.reshape(x.shape[0], -1)                  # (p, k)
    p, k = xm.shape
    kind, shape = classify_laplacian(L, p)
    dev = require_gpu()
    h = torch.from_numpy(np.ascontiguousarray(xm.T, dtype=np.float32)).to(dev)  # (k, p)
    if kind == "identity":
        hl = h
    else:
        hl = torch.empty_like(h)
        _lib.check(_lib.lib.espm_mu_laplacian(_ptr(h), k, shape[0], shape[1], p, _ptr(hl), _stream()))
    t = (h.double() * hl.double())
    return float(t.mean() if average else t.sum())
