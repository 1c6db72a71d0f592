#!/usr/bin/env python3
"""Pieces of the initialisation timed one by one on the device: the fp64 Gram product of the tall factor as rocBLAS runs it (7 ms at
262144 x 15) against the batched form over blocks of rows (0.04 ms: espm_amd/init_device.py::_gram64), the Cholesky QR around it, and the
NNDSVD post-processing's passes."""
import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda", 0)
p, r = 262144, 15
A = torch.rand((p, r), device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
Ad = A.to(torch.float64)
print("to fp64", t(lambda: A.to(torch.float64)))
print("gram fp64 Ad.T @ Ad", t(lambda: Ad.T @ Ad))
A3 = Ad.view(256, 1024, r)
print("gram fp64 bmm chunks 256", t(lambda: torch.bmm(A3.transpose(1, 2), A3).sum(0)))
A3 = Ad.view(2048, 128, r)
print("gram fp64 bmm chunks 2048", t(lambda: torch.bmm(A3.transpose(1, 2), A3).sum(0)))
print("gram via einsum", t(lambda: torch.einsum('pi,pj->ij', Ad, Ad)))
print("gram elementwise (p, r, r) sum", t(lambda: (Ad[:, :, None] * Ad[:, None, :]).sum(0)))
Ri = torch.rand((r, r), device=dev, dtype=torch.float64)
print("Ad @ Rinv", t(lambda: Ad @ Ri))
print("gram .cpu()", t(lambda: (Ad.T @ Ad).cpu()))
G = (Ad.T @ Ad).cpu().numpy()
from scipy.linalg import solve_triangular
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        R = np.linalg.cholesky(G).T
    t1 = time.perf_counter()
    for _ in range(10):
        Rinv = solve_triangular(R, np.eye(r), lower=False)
    print("host chol", (t1 - t0) / 10 * 1e3, "inverse", (time.perf_counter() - t1) / 10 * 1e3)
print("Rinv to device", t(lambda: torch.from_numpy(Rinv).to(dev)))
from espm_amd import init_device as idv
print("_qr_tall whole", t(lambda: idv._qr_tall(A)))
# nndsvd post pieces
k = 5
Vd = torch.randn((k, p), device=dev)
print("clamps", t(lambda: (Vd.clamp_min(0), (-Vd).clamp_min(0))))
Yp, Yn = Vd.clamp_min(0), (-Vd).clamp_min(0)
print("norms + cpu", t(lambda: torch.stack((torch.linalg.vector_norm(Yp, dim=1), torch.linalg.vector_norm(Yn, dim=1))).cpu()))
Hd = Vd.abs()
zh = Hd < 0.5
print("mask count", t(lambda: int(zh.sum())))
cnt = int(zh.sum())
rng = np.random.RandomState(0)
t0 = time.perf_counter(); vals = abs(0.2 * rng.standard_normal(size=cnt) / 100); print("host standard_normal", cnt, (time.perf_counter() - t0) * 1e3)
print("vals to device + masked assign", t(lambda: Hd.__setitem__(zh, torch.from_numpy(vals).to(device=dev, dtype=Hd.dtype))))
print("H cpu", t(lambda: Hd.cpu()))
