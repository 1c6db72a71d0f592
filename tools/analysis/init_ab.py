#!/usr/bin/env python3
"""The fp32 NNDSVD case of tests/test_gpu_estimator.py::test_device_nndsvd_matches_sklearn under the four combinations of LU / QR routes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sklearn.decomposition._nmf import _initialize_nmf
from espm_amd import init_device
n, p, k, init, dtype = 256, 17000, 3, "nndsvdar", np.float32
rng = np.random.default_rng(n)
W = rng.random((n, k)) ** 3
H = rng.random((k, p)) ** 2
X = rng.poisson(3.0 * W @ H).astype(dtype)
Wr, Hr = _initialize_nmf(X, n_components=k, init=init, random_state=7)
print("zeros in the reference's svd-based factors (before the random fill): see counts of filled entries")
for lu in ("hip", "torch"):
    for qr in ("chol", "torch"):
        os.environ["ESPM_INIT_LU"], os.environ["ESPM_INIT_QR"] = lu, qr
        Wd, Hd = init_device.initialize_nmf_device(X, k, init=init, random_state=7)
        bw = np.abs(Wd - Wr) > 2e-4 * np.abs(Wr).max() + 2e-4 * np.abs(Wr)
        bh = np.abs(Hd - Hr) > 2e-4 * np.abs(Hr).max() + 2e-4 * np.abs(Hr)
        print(f"LU {lu:5s} QR {qr:5s}: W mismatches {bw.sum():6d} (max {np.abs(Wd - Wr).max():.3e}), H mismatches {bh.sum():6d} (max {np.abs(Hd - Hr).max():.3e});"
              f" first bad H index {np.argwhere(bh)[0] if bh.any() else None}")
        U, S, V = init_device.randomized_svd_device(__import__('torch').from_numpy(X).cuda(), k, 7)
        from sklearn.utils.extmath import randomized_svd
        Ur, Sr, Vr = randomized_svd(X, k, random_state=7)
        print(f"      singular values {S} vs {Sr}; max |dU| {np.abs(U - Ur).max():.3e}, max |dV| {np.abs(V - Vr).max():.3e}")
