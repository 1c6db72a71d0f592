"""Objective pieces of the path (espm/measures.py:456-504, :524-548, :560-577; KLdiv :387-425) on the GPU, the majorisers the path's
tests check the updates against (KL_loss_surrogate, log_surrogate: host numpy) and the ground-truth comparison of a fit
(espm/measures.py:13-47, :99-119, :125-339, :579-626; k x k problems, host numpy)."""
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


def KLdiv(X, D, H, log_shift=log_shift, average=False):
    """Generalised KL divergence D_KL(X || D H) = sum X log(X / DH) + sum(DH - X), all three clamped at log_shift (espm/measures.py:387-425).

    The factorised form of the path's data term: the H-step kernel in loss-only mode forms exactly this sum per pixel tile (the state's
    record: KL part + sum of D H - sum of X), in fp32 element-wise with fp64 sums."""
    from espm_amd.engine import MUEngine

    X = np.asarray(X)
    D = np.maximum(np.asarray(D), log_shift)
    H = np.maximum(np.asarray(H), log_shift)
    eng = MUEngine(X, H.shape[0], fix_zero_lines=False, max_iter=1, log_shift=log_shift, simplex_W=False)
    eng.load_state(D, H)
    eng.eval_current(advance_h=False)
    val = float(eng.history(average=False)["kl"][0])   # (X = 0 contributes log_shift * log(log_shift / Y) in the reference: below the sum's resolution)
    return val / X.size if average else val


def KL(X, Y, log_shift=log_shift, average=False):
    """Generalised KL divergence of two matrices, sum X log(X / Y) + sum(Y - X) with both clamped at log_shift (espm/measures.py:427-454):
    a reporting measure of arbitrary arrays - element-wise in fp64 with torch where the arrays are, the host for numpy input."""
    import torch

    Xt = torch.as_tensor(np.asarray(X) if not isinstance(X, torch.Tensor) else X).to(torch.float64).clamp_min(log_shift)
    Yt = torch.as_tensor(np.asarray(Y) if not isinstance(Y, torch.Tensor) else Y).to(torch.float64).clamp_min(log_shift)
    red = torch.mean if average else torch.sum
    return float((red(Yt) - red(Xt)) + (red(Xt * torch.log(Xt)) - red(Xt * torch.log(Yt))))


def KL_loss_surrogate(X, W, H, Ht, log_shift=log_shift, average=False):
    """The majoriser of the KL data term at Ht that the multiplicative H update minimises (espm/measures.py:506-522):
    sum_ij X_ij sum_k U_ikj log(U_ikj / (W_ik H_kj)) + sum(W Ht), U_ikj = W_ik Ht_kj / (W Ht)_ij.

    The inner sum over k collapses - log(U / (W H)) = log(Ht / H)_kj - log(W Ht)_ij and sum_k U = 1 - to
    (W (Ht log(Ht / H)))_ij / (W Ht)_ij - log (W Ht)_ij: two small products instead of the reference's n x k x p arrays, the same value to rounding."""
    W = np.maximum(np.asarray(W, dtype=np.float64), log_shift)
    H = np.maximum(np.asarray(H, dtype=np.float64), log_shift)
    Ht = np.maximum(np.asarray(Ht, dtype=np.float64), log_shift)
    X = np.maximum(np.asarray(X, dtype=np.float64), log_shift)
    WHt = W @ Ht
    inner = (W @ (Ht * np.log(Ht / H))) / WHt - np.log(WHt)
    if average:   # (the reference averages X * inner + sum_k W Ht over the n x p entries)
        return float(np.mean(X * inner + WHt))
    return float(np.sum(X * inner) + np.sum(WHt))


def log_surrogate(H, Ht, mu, epsilon, average=False):
    """The tangent majoriser of the log sparsity term at Ht (espm/measures.py:550-558): sum_kj mu_k (log(Ht + eps) + (H - Ht) / (Ht + eps))."""
    H, Ht = np.asarray(H), np.asarray(Ht)
    weight = mu if np.isscalar(mu) else np.asarray(mu)[:, None]
    terms = weight * (np.log(Ht + epsilon) + (H - Ht) / (Ht + epsilon))
    return np.mean(terms) if average else np.sum(terms)


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


# ---- ground-truth comparison (true_D / true_H tracking, espm/estimators/base.py:335-347) --------------------------
def spectral_angle(v1, v2):
    """Angle in degrees between two spectra, or between the rows of two (phases, channels) arrays (espm/measures.py:13-47)."""
    v1, v2 = np.asarray(v1, dtype=np.float64), np.asarray(v2, dtype=np.float64)
    if v1.ndim == 1:
        if v1.shape != v2.shape:
            raise ValueError("v1 and v2 should have the same shape.")
        return np.arccos(np.clip(np.dot(v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)), -1.0, 1.0)) * 180 / np.pi
    if v1.shape[1] != v2.shape[1]:
        raise ValueError("The second dimensions of v1 and v2 should be the same.")
    a = v1 / np.sqrt(np.sum(v1 ** 2, axis=1, keepdims=True))
    b = v2 / np.sqrt(np.sum(v2 ** 2, axis=1, keepdims=True))
    return np.arccos(np.clip(a @ b.T, -1.0, 1.0)) * 180 / np.pi


def squared_distance(x, y=None):
    """Mean squared distance between the rows of x and the rows of y (espm/measures.py:579-626)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    y = x if y is None else np.atleast_2d(np.asarray(y, dtype=np.float64))
    if x.shape[1] != y.shape[1]:
        raise ValueError("The sizes of x and y do not fit")
    xx, yy = (x * x).sum(axis=1), (y * y).sum(axis=1)
    return np.abs(xx[:, None] + yy[None, :] - 2 * np.dot(x, y.T)) / x.shape[1]


def global_min(matr):
    """Row-wise minima and their columns (espm/measures.py:157-170)."""
    import warnings
    res = [float(np.min(row)) for row in matr]
    ind = [int(np.argmin(row)) for row in matr]
    if any(ind.count(x) > 1 for x in ind):
        warnings.warn("Several results share the same truth")
    return res, ind


def unique_min(matrix):
    """The one-row-per-column assignment of a square matrix with the smallest sum, by brute force over the permutations
    (espm/measures.py:172-207; not meant for more than ~10 phases): its entries in column order and the permutation."""
    from itertools import permutations
    matrix = np.asarray(matrix)
    k = matrix.shape[0]
    perms = list(permutations(range(k), k))
    sums = [sum(matrix[perm[i], i] for i in range(k)) for perm in perms]
    best = perms[sums.index(min(sums))]
    return [matrix[best[i], i] for i in range(k)], best


def find_min_angle(true_vectors, algo_vectors, get_ind=False, unique=False):
    """Best match of NMF spectra to true spectra by spectral angle (espm/measures.py:125-155)."""
    out = (unique_min if unique else global_min)(spectral_angle(true_vectors, algo_vectors))
    return out if get_ind else out[0]


def find_min_MSE(true_maps, algo_maps, get_ind=False, unique=False):
    """Best match of NMF maps to true maps by mean squared error (espm/measures.py:244-270)."""
    out = (unique_min if unique else global_min)(squared_distance(true_maps, algo_maps))
    return out if get_ind else out[0]


def mse(map1, map2):
    """Mean squared error between two maps (espm/measures.py:51-72)."""
    return np.mean((np.asarray(map1) - np.asarray(map2)) ** 2)


def mae(map1, map2):
    """Mean absolute error between two maps (espm/measures.py:75-96)."""
    return np.mean(np.abs(np.asarray(map1) - np.asarray(map2)))


def r2(map_true, map_pred):
    """Coefficient of determination between two maps, every leading index an output (espm/measures.py:99-119: scikit-learn's r2_score
    on the maps reshaped to (first axis, rest))."""
    from sklearn.metrics import r2_score
    a, b = np.asarray(map_true), np.asarray(map_pred)
    return r2_score(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1))


def ordered_r2(true_maps, algo_maps, input_inds):
    """R^2 of every phase for a given correspondence (espm/measures.py:315-327)."""
    return [float(r2(true_maps[j], algo_maps[i])) for i, j in enumerate(input_inds)]


def ordered_mse(true_maps, algo_maps, input_inds):
    """MSE of every phase for a given correspondence (espm/measures.py:287-299)."""
    return [float(mse(true_maps[j], algo_maps[i])) for i, j in enumerate(input_inds)]


def ordered_mae(true_maps, algo_maps, input_inds):
    """MAE of every phase for a given correspondence (espm/measures.py:301-313)."""
    return [float(mae(true_maps[j], algo_maps[i])) for i, j in enumerate(input_inds)]


def ordered_angles(true_spectra, algo_spectra, input_inds):
    """Spectral angle of every phase for a given correspondence (espm/measures.py:330-339)."""
    return [spectral_angle(true_spectra[j], algo_spectra[i]) for i, j in enumerate(input_inds)]


def find_min_config(true_maps, true_spectra, algo_maps, algo_spectra, angles=True):
    """Best match of NMF phases to the truth by angles (or by map errors) and whether the two criteria agree
    (espm/measures.py:222-256)."""
    min_MSE_config = find_min_MSE(true_maps, algo_maps, get_ind=True, unique=True)[1]
    min_angle_config = find_min_angle(true_spectra, algo_spectra, get_ind=True, unique=True)[1]
    warning = False
    if min_MSE_config != min_angle_config:
        print("WARNING : angles and mse disagree there's probably an issue")
        warning = True
    if angles:
        return (find_min_angle(true_spectra, algo_spectra, unique=True), ordered_mse(true_maps, algo_maps, min_angle_config),
                min_angle_config, warning)
    return (ordered_angles(true_spectra, algo_spectra, min_MSE_config), find_min_MSE(true_maps, algo_maps, unique=True),
            min_MSE_config, warning)


def Frobenius_loss(X, W, H, average=False):
    """||X - W H||_F^2 (espm/measures.py:350-384): one GEMM and an element-wise reduction with torch on the device."""
    import torch

    from espm_amd.engine import require_gpu
    dev = require_gpu()
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    r = Xd - torch.from_numpy(np.ascontiguousarray(W, dtype=np.float64)).to(dev) @ torch.from_numpy(np.ascontiguousarray(H, dtype=np.float64)).to(dev)
    t = r * r
    return float(t.mean() if average else t.sum())
