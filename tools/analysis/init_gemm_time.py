#!/usr/bin/env python3
"""The tall-skinny products of the randomized SVD at the headline size through torch (rocBLAS / hipBLASLt): X (2048 x 262144 fp32),
Q with 15 columns - time per product against the 2.1 GB they read."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
dev = torch.device("cuda", 0)
n, p, r = 2048, 512 * 512, 15
X = torch.rand((n, p), device=dev)
M = X.T                         # (p, n) view, as init_device.randomized_svd_device uses it
Q1 = torch.rand((n, r), device=dev)
Q2 = torch.rand((p, r), device=dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, fn in (("M @ Q    (p x n)(n x 15): X^T Q", lambda: M @ Q1), ("M.T @ Q  (n x p)(p x 15): X Q", lambda: M.T @ Q2),
                 ("Q.T @ M  (15 x p)(p x n)", lambda: Q2.T @ M), ("X.sum()", lambda: X.sum())):
    ms = t(fn)
    print(f"{name:34s}: {ms:7.3f} ms  = {X.numel() * 4 / ms / 1e9:6.2f} TB/s of X")
from espm_amd.init_device import _lu_pl
A = torch.rand((p, r), device=dev)
print(f"_lu_pl (p x 15): {t(lambda: _lu_pl(A)):7.3f} ms;  (n x 15): {t(lambda: _lu_pl(Q1)):7.3f} ms;  qr (p x 15): {t(lambda: torch.linalg.qr(A, mode='reduced')):7.3f} ms")

# ---- the sections of initialize_nmf_device at this size (synchronised after each) ----
from espm_amd import init_device as idv
from scipy import linalg
from sklearn.utils import check_random_state
Xc = torch.poisson(torch.full((n, p), 0.25, device=dev))


def stamp(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"   {label:46s} {1e3 * (t1 - t0):7.2f} ms")
    return t1


for rep in range(2):
    print("initialize_nmf_device, sections:")
    torch.cuda.synchronize(); t0 = time.perf_counter(); t_all = t0
    rs = check_random_state(0)
    Mx = Xc.T
    Q = torch.from_numpy(rs.normal(size=(Mx.shape[1], 15))).to(device=dev, dtype=Xc.dtype)
    t0 = stamp("random test matrix -> device", t0)
    for _ in range(7):
        Q = idv._lu_pl(Mx @ Q)
        Q = idv._lu_pl(Mx.T @ Q)
    t0 = stamp("7 power iterations (2 products + 2 LU each)", t0)
    Q = idv._qr_tall(Mx @ Q)
    t0 = stamp("product + Cholesky QR twice (p x 15)", t0)
    B = (Q.T @ Mx).cpu().numpy()
    t0 = stamp("B = Q^T M -> host", t0)
    Uhat, sv, Vt = linalg.svd(B, full_matrices=False, lapack_driver="gesdd")
    t0 = stamp("host SVD of B (15 x n)", t0)
    U = (Q @ torch.from_numpy(Uhat).to(device=dev, dtype=Xc.dtype)).cpu().numpy()
    t0 = stamp("U = Q Uhat -> host (p x 15)", t0)
    avg = float(Xc.mean(dtype=torch.float64))
    t0 = stamp("mean of X (fp64)", t0)
