"""Host-side helpers of the path (espm/utils.py:15-96)."""
import numpy as np
import scipy.sparse as sp


def process_losses(losses):
    """Structured loss array -> (values, names), espm/utils.py:15-37."""
    names = losses.dtype.names
    values = np.array([[row[i] for row in losses] for i in range(len(names))])
    return values, names


def create_laplacian_matrix(nx, ny=None):
    """Sparse (p, p) 5-point graph Laplacian with zero-flux boundary (espm/utils.py:39-76).

    The device kernels apply this operator as a stencil (they never read the matrix); the matrix
    is provided because it is part of the reference's public surface (``est.L_``, tests)."""
    if ny is None:
        ny = nx
    assert nx > 1
    assert ny > 1
    idx = np.arange(nx * ny).reshape(nx, ny)
    src = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    dst = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    rows, cols = np.concatenate([src, dst]), np.concatenate([dst, src])
    adj = sp.coo_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(nx * ny, nx * ny)).tocsr()
    deg = sp.diags(np.asarray(adj.sum(axis=1)).ravel().astype(np.float32))
    return (deg - adj).tocsr()


def identity_laplacian(p):
    """``L_`` when ``shape_2d`` is None (espm/estimators/base.py:289-291)."""
    return sp.identity(p, dtype=np.float32, format="csr")


def classify_laplacian(L, p):
    """Recognise the operators the device stencil implements.

    Returns ("identity", None) or ("grid", (nx, ny)); raises NotImplementedError for any other
    matrix (the HIP path has no generic sparse product and there is no CPU fallback)."""
    L = sp.csr_matrix(L)
    if L.shape != (p, p):
        raise ValueError(f"L must be ({p}, {p}), got {L.shape}")
    if (L - identity_laplacian(p)).count_nonzero() == 0:
        return "identity", None
    coo = L.tocoo()
    off = np.abs(coo.row - coo.col)
    ny = int(off.max()) if off.size else 0
    if ny > 0 and p % ny == 0 and p // ny > 1 and ny > 1:
        nx = p // ny
        if abs(L - create_laplacian_matrix(nx, ny)).max() == 0:
            return "grid", (nx, ny)
    raise NotImplementedError("only the identity and the 2-D grid Laplacian of create_laplacian_matrix are "
                              "supported by the device stencil")


def rescaled_DH(D, H):
    """Rescale so that the columns of H sum approximately to one (espm/utils.py:79-96)."""
    from scipy.optimize import nnls

    o = np.ones((H.shape[1],))
    s = np.linalg.lstsq(H.T, o, rcond=None)[0]
    if (s <= 0).any():
        s = np.maximum(nnls(H.T, o)[0], 1e-10)
    return D @ np.diag(1 / s), np.diag(s) @ H
