"""NNDSVD initialisation with the randomized SVD's passes over X on the GPU.

The reference initialises through scikit-learn's ``_initialize_nmf`` (espm/estimators/updates.py:3, :179), whose
randomized SVD makes 2 * n_iter + 2 = 16 passes over X on the host: at the headline size (2048 x 262144, fp64)
that takes far longer than the whole multiplicative-update loop on the GPU.  This module follows the same
algorithm step by step (sklearn.utils.extmath._randomized_svd / _randomized_range_finder, scikit-learn 1.7:
Gaussian test matrix from the same ``RandomState``, LU-normalised power iterations, QR, SVD of the small
projection, ``svd_flip``; then sklearn.decomposition._nmf._initialize_nmf's NNDSVD post-processing) and moves only
the products with X and the tall-skinny LU / QR factorisations to the device (torch.linalg: plumbing, like the
rest of the host side).  Everything small - the random matrix, the SVD of the (k + 10) x n projection, the
NNDSVD sign logic - stays in numpy / scipy in fp64, so the result equals scikit-learn's to rounding
(tests/test_gpu_estimator.py::test_device_nndsvd_matches_sklearn).
"""
from __future__ import annotations

import os

import numpy as np
import torch

# X with at least this many entries is initialised on the device (smaller ones are instantaneous on the host)
DEVICE_INIT_MIN_SIZE = 4_000_000


def _lu_pl(A):
    """P @ L of A = P L U (scipy.linalg.lu(A, permute_l=True)[0]) for a tall matrix on the device.

    Right-looking elimination with partial pivoting over the k (= n_components + 10) columns, the pivot chosen on the
    device (no read-back): per column a masked arg-max, the multipliers, one rank-1 update of the remaining columns - a few
    passes over a (rows x k) matrix.  torch.linalg.lu_factor took 6 ms per call on the 262144 x 15 matrices of the headline
    size (14 calls: half of the initialisation); this takes 1 ms.  Row i of P L holds the multipliers of the steps before row
    i became a pivot, 1 at its own step, 0 after; rows never chosen hold all k multipliers.  Same pivots as LAPACK's getrf
    (first row of the largest modulus), hence the same factors up to the rounding of a different order of operations.

    On the device this is espm_lu_pl of libespm_mu (csrc/mu_init.hip: ONE launch per column, 0.1 ms instead of the 1.35 ms the
    torch operations below take at any height - launch latency); the torch formulation remains for host tensors (the CPU tests of
    the algorithm against scipy) and is what the kernel's arithmetic follows."""
    if A.is_cuda and A.shape[0] >= A.shape[1] and A.shape[1] <= 64 and os.environ.get("ESPM_INIT_LU") != "torch":   # (a wide matrix - fewer channels than columns - takes the torch route)
        return _lu_pl_device(A)
    m, k = A.shape
    r = min(m, k)
    M = A.clone()
    out = torch.zeros((m, r), dtype=A.dtype, device=A.device)
    free = torch.ones(m, dtype=torch.bool, device=A.device)     # rows that have not been a pivot yet
    zero = torch.zeros((), dtype=A.dtype, device=A.device)
    for j in range(r):
        col = M[:, j]
        i = torch.argmax(torch.where(free, col.abs(), -1.0)).view(1)
        prow = M.index_select(0, i)                             # (1, k): row j of U from column j on
        pv = prow[0, j]
        mult = torch.where(free & (pv != 0), col / pv, zero)    # (a zero pivot: the column is zero below, LAPACK leaves zeros)
        mult.index_fill_(0, i, 1.0)
        out[:, j] = mult
        free.index_fill_(0, i, False)
        if j + 1 < k:
            M[:, j + 1:] -= torch.where(free, mult, zero)[:, None] * prow[:, j + 1:]
    return out


def _lu_pl_device(A):
    import ctypes as C
    from . import _lib
    if A.dtype not in (torch.float32, torch.float64) or A.dim() != 2 or A.shape[0] < A.shape[1] or A.shape[1] > 64:
        raise ValueError(f"_lu_pl: a tall fp32 / fp64 matrix of at most 64 columns is expected, got {tuple(A.shape)} {A.dtype}")
    if A.stride(1) != 1:
        A = A.contiguous()
    m, r = A.shape
    dt = _lib.SRC_F64 if A.dtype == torch.float64 else _lib.SRC_F32
    nbytes = int(_lib.lib.espm_lu_pl_scratch_bytes(m, r, dt))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    out = torch.empty((m, r), dtype=A.dtype, device=A.device)
    with torch.cuda.device(A.device):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib.espm_lu_pl(C.c_void_p(A.data_ptr()), dt, m, r, A.stride(0), C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                       nbytes, stream))
    return out


def _gram64(Ad):
    """Ad^T Ad of a tall fp64 matrix (m >> r) as a batched product over blocks of 1024 rows, summed: rocBLAS gives the plain
    `Ad.T @ Ad` of a 262144 x 15 matrix ONE workgroup's worth of parallelism - 7.0 ms, against 0.04 ms for the blocks
    (profiles/r05z_qr_bench.log); two of them were half of the initialisation's device time."""
    m, r = Ad.shape
    blk = 1024
    main = (m // blk) * blk
    if main < 4 * blk:
        return Ad.T @ Ad
    A3 = Ad[:main].reshape(main // blk, blk, r)
    G = torch.bmm(A3.transpose(1, 2), A3).sum(dim=0)
    if main < m:
        G = G + Ad[main:].T @ Ad[main:]
    return G


def _qr_tall(A):
    """Q of the reduced QR factorisation of a tall matrix (m >> r columns), up to the signs of its columns - which the randomized
    SVD does not see: U = Q Uhat with Uhat from the SVD of Q^T M is the same for Q and Q D, D = diag(+-1).

    torch.linalg.qr takes 10 ms on the 262144 x 15 matrix of the headline size (Householder reflections, a launch per column and
    more); here: Cholesky QR, twice (A = Q1 R1 with R1 = chol(A^T A), then the same on Q1, whose Gram matrix is the identity to
    cond(A)^2 eps: the second pass restores orthonormality to rounding - Yamamoto et al., "Roundoff error analysis of the
    CholeskyQR2 algorithm", ETNA 44, 2015), Gram matrices, factors and products in fp64 whatever the dtype of A: two r x r
    Gram products over A (_gram64), two r x r factorisations on the host and two products with R^-1, under 1 ms.  The triangular factors keep the nesting of the column spaces, so this is
    THE QR factor up to signs, not just some orthonormal basis.  A Gram matrix that is not numerically positive definite
    (rank-deficient A: fewer independent directions in X than n_components + 10) goes to torch.linalg.qr."""
    Ad = A.to(torch.float64)
    for _ in range(2):
        # the r x r factor on the host (numpy / scipy, fp64): the device only forms the Gram matrix and multiplies by R^-1 - a
        # triangular solve with 262144 right-hand rows takes rocBLAS 60 ms, the product with the explicit inverse 0.1
        G = _gram64(Ad).cpu().numpy()
        try:
            R = np.linalg.cholesky(G).T                      # upper factor: G = R^T R
        except np.linalg.LinAlgError:
            return torch.linalg.qr(A, mode="reduced")[0]
        d = np.diagonal(R)
        if not np.isfinite(R).all() or d.min() <= 1e-7 * d.max():      # cond(A) beyond what two passes repair: the Householder route
            return torch.linalg.qr(A, mode="reduced")[0]
        from scipy.linalg import solve_triangular
        Rinv = solve_triangular(R, np.eye(R.shape[0]), lower=False)
        Ad = Ad @ torch.from_numpy(Rinv).to(Ad.device)
    return Ad.to(A.dtype)


def randomized_svd_device(Xd, n_components, random_state, n_oversamples=10, long_on_device=False):
    """sklearn.utils.extmath._randomized_svd(M, n_components, random_state=...) with its defaults, M on the device.

    Returns numpy (U (n_samples, k), s (k), Vt (k, n_features)).  ``long_on_device`` (more features than samples - an image's
    pixels against its channels): Vt comes back as a DEVICE tensor (k, n_features), signs applied - the NNDSVD's passes over it run
    there (initialize_nmf_device) and 15 floats per pixel stay off the PCIe link."""
    from scipy import linalg
    from sklearn.utils import check_random_state

    rs = check_random_state(random_state)
    n_random = n_components + n_oversamples
    n_samples, n_features = Xd.shape
    n_iter = 7 if n_components < 0.1 * min(Xd.shape) else 4
    transpose = n_samples < n_features
    M = Xd.T if transpose else Xd
    Q = torch.from_numpy(rs.normal(size=(M.shape[1], n_random))).to(device=Xd.device, dtype=Xd.dtype)
    normalize = _lu_pl if n_iter > 2 else (lambda A: A)  # power_iteration_normalizer="auto"
    for _ in range(n_iter):
        Q = normalize(M @ Q)
        Q = normalize(M.T @ Q)
    Q = _qr_tall(M @ Q) if os.environ.get("ESPM_INIT_QR") != "torch" else torch.linalg.qr(M @ Q, mode="reduced")[0]
    B = (Q.T @ M).cpu().numpy()
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver="gesdd")
    if transpose and long_on_device:
        # svd_flip with u_based_decision=False: the signs come from the rows of the SHORT factor, which is on the host anyway
        idx = np.argmax(np.abs(Vt), axis=1)
        signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
        Vt *= signs[:, np.newaxis]
        Ud = Q @ torch.from_numpy(Uhat[:, :n_components] * signs[np.newaxis, :n_components]).to(device=Xd.device, dtype=Xd.dtype)   # (p, k)
        return Vt[:n_components, :].T, s[:n_components], Ud.T.contiguous()
    U = (Q @ torch.from_numpy(Uhat).to(device=Xd.device, dtype=Xd.dtype)).cpu().numpy()
    # svd_flip: signs from the rows of Vt when transposed (u_based_decision=False), from the columns of U otherwise
    if transpose:
        idx = np.argmax(np.abs(Vt), axis=1)
        signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    else:
        idx = np.argmax(np.abs(U), axis=0)
        signs = np.sign(U[idx, np.arange(U.shape[1])])
    U *= signs[np.newaxis, :]
    Vt *= signs[:, np.newaxis]
    if transpose:
        return Vt[:n_components, :].T, s[:n_components], U[:, :n_components].T
    return U[:, :n_components], s[:n_components], Vt[:n_components, :]


def _cholqr_rows_sharded(A, group, passes):
    """Orthonormal basis of the column space of a tall matrix whose ROWS are spread over the ranks of ``group`` (A: this rank's
    rows): Cholesky QR - the r x r Gram matrix added over the ranks (fp64), its factor on the host, A R^-1 - once as the
    normaliser of a power iteration, twice for the final basis (_qr_tall's argument).  Every rank factors the same all-reduced
    Gram matrix: the same R everywhere.  A Gram matrix that is not numerically positive definite gets a relative ridge (the
    basis is then orthonormal only to that, which the second pass repairs or the SVD of the projection absorbs)."""
    from scipy.linalg import solve_triangular
    Ad = A.to(torch.float64)
    for _ in range(passes):
        G = _gram64(Ad)
        torch.distributed.all_reduce(G, group=group)
        Gh = G.cpu().numpy()
        Gh = 0.5 * (Gh + Gh.T)
        ridge = 0.0
        for _try in range(8):
            try:
                R = np.linalg.cholesky(Gh + ridge * np.eye(Gh.shape[0])).T
                if np.isfinite(R).all() and np.diagonal(R).min() > 0:
                    break
            except np.linalg.LinAlgError:
                pass
            ridge = max(ridge * 100.0, 1e-14 * float(np.trace(Gh)) / Gh.shape[0])
        Rinv = solve_triangular(R, np.eye(R.shape[0]), lower=False)
        Ad = Ad @ torch.from_numpy(Rinv).to(Ad.device)
    return Ad.to(A.dtype)


def randomized_svd_sharded(Xd, n_components, random_state, shard, n_oversamples=10):
    """``randomized_svd_device`` for an image whose PIXELS are spread over the ranks: Xd is this rank's (n_samples = channels,
    block of pixels) matrix, ``shard`` the estimator's _Shard.  The algorithm is the same - Gaussian test matrix from the same
    ``RandomState`` (replicated: every rank draws it), power iterations, QR, SVD of the small projection, ``svd_flip`` - with the
    contractions over the pixels added over the ranks (n x r and r x r matrices: ~100 KB per all-reduce) and nothing of the
    size of the image ever leaving its rank.  One deviation from scikit-learn: the power iterations normalise the TALL factor
    (pixels x r, its rows on different ranks) by Cholesky QR instead of by the L of a partially pivoted LU, whose pivot search
    would be a collective per column; both span the same column space, which is all a normaliser of the iteration is asked for,
    so U, s, V agree with the one-GPU routine to rounding (tests/test_gpu_sharded_estimator.py).  The channel-side factor (n x r,
    replicated) keeps the LU.  Returns numpy (U (n, k) replicated, s (k), Vt (k, p) assembled from the ranks' blocks)."""
    from scipy import linalg
    from sklearn.utils import check_random_state

    group = shard.group
    rs = check_random_state(random_state)
    n_random = n_components + n_oversamples
    n_samples = Xd.shape[0]
    p_total = sum(shard.counts)
    if n_samples >= p_total:
        raise NotImplementedError("sharded initialisation: more channels than pixels")   # (the caller falls back on the whole image)
    n_iter = 7 if n_components < 0.1 * min(n_samples, p_total) else 4
    # (transpose = True in scikit-learn's routine: M = X^T, pixels x channels; Q starts on the channel side)
    Q = torch.from_numpy(rs.normal(size=(n_samples, n_random))).to(device=Xd.device, dtype=Xd.dtype)
    lu = _lu_pl if n_iter > 2 else (lambda A: A)
    for _ in range(n_iter):
        Y = Xd.T @ Q                                            # (pixels of this rank, r)
        if n_iter > 2:
            Y = _cholqr_rows_sharded(Y, group, 1)
        Z = Xd @ Y                                              # (n, r): a sum over the pixels
        torch.distributed.all_reduce(Z, group=group)
        Q = lu(Z)
    Y = _cholqr_rows_sharded(Xd.T @ Q, group, 2)
    B = Y.T @ Xd.T                                              # (r, n): a sum over the pixels
    torch.distributed.all_reduce(B, group=group)
    Uhat, s, Vt = linalg.svd(B.cpu().numpy(), full_matrices=False, lapack_driver="gesdd")
    U_loc = Y @ torch.from_numpy(Uhat).to(device=Xd.device, dtype=Xd.dtype)     # this rank's rows of the pixel-side factor
    idx = np.argmax(np.abs(Vt), axis=1)                                         # svd_flip, u_based_decision=False
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    Vt *= signs[:, np.newaxis]
    U_loc = U_loc * torch.from_numpy(signs).to(device=Xd.device, dtype=Xd.dtype)[None, :]
    V_full = shard.gather_cols(U_loc[:, :n_components].T.contiguous())          # (k, p): every rank's block, in rank order
    return Vt[:n_components, :].T, s[:n_components], V_full


_TORCH_OF = {"float32": torch.float32, "float64": torch.float64}


def _nndsvd_long_on_device(U, S, Vd, n_components, init, random_state, eps, avg):
    """The NNDSVD post-processing of sklearn.decomposition._nmf._initialize_nmf (the loop over the singular triplets, the zeroing
    below eps, the fill of the zeros) with the LONG factor - Vd (k, pixels) - on the device: at 5 x 262144 its ~15 numpy passes per
    component were 17 of the initialisation's 36 ms on the host (profiles/r05f_init_profile.log).  The same operations in the same
    order and dtype; what differs from the host route is the summation order inside the four norms (1e-7 relative).  The short factor
    U (channels, k) stays in numpy.  Returns numpy (W, H)."""
    from sklearn.utils import check_random_state
    dev, dt = Vd.device, Vd.dtype
    k = n_components
    Yp, Yn = Vd.clamp_min(0), (-Vd).clamp_min(0)            # max(y, 0), |min(y, 0)|
    nrm = torch.stack((torch.linalg.vector_norm(Yp, dim=1), torch.linalg.vector_norm(Yn, dim=1))).cpu().numpy()   # ONE read-back
    W = np.zeros_like(U)
    Hd = torch.zeros_like(Vd)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    Hd[0] = float(np.sqrt(S[0])) * Vd[0].abs()
    for j in range(1, k):
        x = U[:, j]
        x_p, x_n = np.maximum(x, 0), np.abs(np.minimum(x, 0))
        x_p_nrm, x_n_nrm = np.linalg.norm(x_p), np.linalg.norm(x_n)
        y_p_nrm, y_n_nrm = nrm[0, j].astype(U.dtype), nrm[1, j].astype(U.dtype)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            u, vd, vn, sigma = x_p / x_p_nrm, Yp[j], y_p_nrm, m_p
        else:
            u, vd, vn, sigma = x_n / x_n_nrm, Yn[j], y_n_nrm, m_n
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u
        Hd[j] = float(lbd) * (vd / float(vn))
    del Yp, Yn
    zw, zh = W < eps, Hd < eps
    W[zw] = 0
    Hd[zh] = 0
    if init == "nndsvda":
        W[zw] = avg
        Hd[zh] = avg
    elif init == "nndsvdar":
        rng = check_random_state(random_state)
        W[zw] = abs(avg * rng.standard_normal(size=int(zw.sum())) / 100)
        vals = abs(avg * rng.standard_normal(size=int(zh.sum())) / 100)       # (the same stream, W's draw first)
        Hd[zh] = torch.from_numpy(np.asarray(vals)).to(device=dev, dtype=dt)  # (boolean-mask assignment fills in C order on both sides)
    return W, Hd.cpu().numpy()


def initialize_nmf_device(X, n_components, init=None, random_state=None, eps=1e-6, device=None, X_device=None, X_mean=None, shard=None):
    """sklearn.decomposition._nmf._initialize_nmf for the NNDSVD family with the passes over X on the GPU.

    X: (n_samples, n_features) numpy array, fp32 or fp64 (kept in its dtype, like scikit-learn); X_device: the same
    matrix already on the GPU (then X is only consulted for its shape and dtype); X_mean: its mean, if the caller has it.
    shard (the estimator's _Shard): X_device holds this rank's block of pixels only - the randomized SVD runs sharded
    (randomized_svd_sharded), its small factors are replicated and the post-processing below is the same on every rank."""
    from sklearn.utils import check_random_state

    if X_device is None and (X < 0).any():
        raise ValueError("Negative values in data passed to NMF initialization")
    n_samples, n_features = X.shape
    if init is None:
        init = "nndsvda"
    if init not in ("nndsvd", "nndsvda", "nndsvdar"):
        raise ValueError(f"initialize_nmf_device handles the NNDSVD family, got init={init!r}")
    if n_components > min(n_samples, n_features):
        raise ValueError("init = '{}' can only be used when n_components <= min(n_samples, n_features)".format(init))
    if X_device is not None:
        Xd = X_device
    else:
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        Xd = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
    if shard is not None:   # X_device is this rank's block of pixels (and X_mean the image's mean)
        if X_device is None or (X_mean is None and init != "nndsvd"):
            raise ValueError("sharded initialisation needs the device block and the mean of the image")
        U, S, V = randomized_svd_sharded(Xd, n_components, random_state, shard)
    else:
        U, S, V = randomized_svd_device(Xd, n_components, random_state, long_on_device=os.environ.get("ESPM_INIT_NNDSVD") != "host")
    # (X_mean: the caller knows the mean of X_device already - one pass over X less)
    avg = (float(X_mean) if X_mean is not None else float(Xd.mean(dtype=torch.float64))) if init != "nndsvd" else 0.0
    del Xd
    if isinstance(V, torch.Tensor):
        return _nndsvd_long_on_device(U.astype(X.dtype, copy=False), S.astype(X.dtype, copy=False), V.to(_TORCH_OF[np.dtype(X.dtype).name]), n_components, init,
                                      random_state, eps, avg)
    U, S, V = U.astype(X.dtype, copy=False), S.astype(X.dtype, copy=False), V.astype(X.dtype, copy=False)
    # (the randomized SVD hands back transposed views: rows of V 15 floats apart in memory - every pass below over a row of 262144
    #  entries would touch a cache line per entry)
    U, V = np.ascontiguousarray(U), np.ascontiguousarray(V)
    W = np.zeros_like(U)
    H = np.zeros_like(V)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    for j in range(1, n_components):
        x, y = U[:, j], V[j, :]
        x_p, y_p = np.maximum(x, 0), np.maximum(y, 0)
        x_n, y_n = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        x_p_nrm, y_p_nrm = np.linalg.norm(x_p), np.linalg.norm(y_p)
        x_n_nrm, y_n_nrm = np.linalg.norm(x_n), np.linalg.norm(y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            u, v, sigma = x_p / x_p_nrm, y_p / y_p_nrm, m_p
        else:
            u, v, sigma = x_n / x_n_nrm, y_n / y_n_nrm, m_n
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u
        H[j, :] = lbd * v
    # (scikit-learn's lines, with every mask formed once: at 5 x 262144 the three passes per `H[H == 0]` add up to milliseconds;
    #  what is below eps IS what is zero afterwards - the factors are non-negative - and boolean-mask assignment fills in C order
    #  either way, so the random stream lands on the same entries)
    zw, zh = W < eps, H < eps
    W[zw] = 0
    H[zh] = 0
    if init == "nndsvda":
        W[zw] = avg
        H[zh] = avg
    elif init == "nndsvdar":
        rng = check_random_state(random_state)
        W[zw] = abs(avg * rng.standard_normal(size=int(zw.sum())) / 100)
        H[zh] = abs(avg * rng.standard_normal(size=int(zh.sum())) / 100)
    return W, H
