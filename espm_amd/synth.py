"""Seeded synthetic spectrum images for tests and the benchmark.

The recipe follows the reference's simulators without using any of their code paths that need
hyperspy: phases come from the ``ToyModel`` construction (espm/models/base.py:78-136: a random
dictionary G of Gaussian peaks, phases = G @ |Laplace|), weights are smooth non-negative maps
summing to one per pixel (required by espm/datasets/base.py:177), and the data are
``X ~ Poisson(N * weights @ (phases * densities))`` (espm/datasets/base.py:56-68).
"""
from __future__ import annotations

import numpy as np


def toy_dictionary(n, C=15, seed=0, n_el=45):
    """(n, C) dictionary of Gaussian peaks, recipe of espm/models/base.py:95-118."""
    rng = np.random.RandomState(seed)
    n_gauss = rng.randint(2, 5, [C])
    ell = np.arange(0, 1, 1 / n)[:n]
    mu_g = rng.rand(n_el)
    sig_g = 1 / n_el + np.abs(rng.randn(n_el)) / n_el / 5
    G = np.zeros((n, C))
    for i, c in enumerate(n_gauss):
        for ind in rng.choice(n_el, size=[c], replace=False):
            w = 0.1 + 0.9 * rng.rand()
            G[:, i] += w * np.exp(-(ell - mu_g[ind]) ** 2 / (2 * sig_g[ind] ** 2))
    return G


def toy_phases(n, k, C=15, seed=0, background=0.02):
    """(k, n) spectra with rows summing to 1 (espm/models/base.py:120-136, datasets/base.py:58)."""
    G = toy_dictionary(n, C, seed)
    rng = np.random.RandomState(seed + 1)
    Wdot = np.abs(rng.laplace(size=[C, k]))
    phases = (G @ Wdot).T + background * np.linspace(1.0, 0.2, n)[None, :]  # smooth continuum
    return phases / phases.sum(axis=1, keepdims=True), G


def smooth_weights(nx, ny, k, seed=0, row0=0, nx_total=None):
    """(nx, ny, k) smooth non-negative maps, sum_k = 1 per pixel.

    Sum of a few separable raised-cosine blobs per component; ``row0`` / ``nx_total`` give the
    rows of a larger image (so that ranks generate consistent shards without communication).
    """
    nx_total = nx if nx_total is None else nx_total
    rng = np.random.RandomState(seed + 7)
    yy = (np.arange(row0, row0 + nx) + 0.5) / nx_total
    xx = (np.arange(ny) + 0.5) / ny
    maps = np.zeros((nx, ny, k))
    for c in range(k):
        acc = np.full((nx, ny), 0.05)
        for _ in range(3):
            cy, cx = rng.rand(2)
            ry, rx = 0.15 + 0.35 * rng.rand(2)
            by = np.clip(1 - np.abs(yy - cy) / ry, 0, None)
            bx = np.clip(1 - np.abs(xx - cx) / rx, 0, None)
            acc += np.outer(0.5 - 0.5 * np.cos(np.pi * by), 0.5 - 0.5 * np.cos(np.pi * bx))
        maps[:, :, c] = acc
    return maps / maps.sum(axis=2, keepdims=True)


def make_problem(n, nx, ny, k, N=500.0, seed=0, m=None, row0=0, nx_total=None):
    """Ground truth of a synthetic spectrum image.

    Returns dict(phases (k, n), weights (nx*ny, k), rates = N * weights @ phases (p, n) float32,
    G (n, m) or None): ``m`` columns = the 15 dictionary columns + (m - 15) smooth background columns.
    """
    phases, Gd = toy_phases(n, k, seed=seed)
    w = smooth_weights(nx, ny, k, seed=seed, row0=row0, nx_total=nx_total).reshape(nx * ny, k)
    G = None
    if m is not None:
        extra = max(m - Gd.shape[1], 0)
        ell = np.linspace(0, 1, n)
        cols = [Gd[:, :min(m, Gd.shape[1])]] + [np.exp(-(1 + 2 * i) * ell)[:, None] for i in range(extra)]
        G = np.concatenate(cols, axis=1) + 1e-6
    return dict(phases=phases, weights=w, N=float(N), G=G, shape_2d=(nx, ny))


def sample_numpy(prob, seed=0):
    """X (n, p) float64 Poisson counts on the host (small problems)."""
    rng = np.random.RandomState(seed)
    rates = prob["N"] * (prob["weights"] @ prob["phases"])
    return rng.poisson(rates).T.astype(np.float64)


def sample_torch(prob, device, seed=0, row0=0, chunk_rows=64):
    """X (p, n) float32 Poisson counts generated on the device, pixel-major (hyperspy's layout).

    The image is drawn in blocks of ``chunk_rows`` image rows, each from its own seed
    ``seed + global block index``: a rank that owns rows [row0, row0 + nx) of a larger image gets
    exactly the pixels a single-GPU run would have there (row0 must be a multiple of chunk_rows).
    """
    import torch

    nx, ny = prob["shape_2d"]
    assert row0 % chunk_rows == 0
    w = torch.from_numpy(prob["weights"].astype(np.float32)).to(device)
    ph = torch.from_numpy((prob["N"] * prob["phases"]).astype(np.float32)).to(device)
    out = torch.empty((w.shape[0], ph.shape[1]), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for r in range(0, nx, chunk_rows):
        g.manual_seed(seed + (row0 + r) // chunk_rows)
        lo, hi = r * ny, min(nx, r + chunk_rows) * ny
        out[lo:hi] = torch.poisson(w[lo:hi] @ ph, generator=g)
    return out


def random_init(n_rows, k, p, seed=0, scale=1.0):
    """W0 ~ U(0,1) * scale, H0 ~ U(0,1) column-normalised."""
    rng = np.random.RandomState(seed + 13)
    W0 = (rng.rand(n_rows, k) + 1e-3) * scale
    H0 = rng.rand(k, p) + 0.05
    H0 /= H0.sum(axis=0, keepdims=True)
    return W0, H0
