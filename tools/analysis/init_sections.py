#!/usr/bin/env python3
"""Where the NNDSVD initialisation's time goes at the headline size (2048 channels x 512 x 512 pixels, k = 5, fp32 counts on the device):
cumulative, synchronised time per kind of step of espm_amd.init_device.initialize_nmf_device - the products with X, the LU
normalisers, the Cholesky QR, host work - by wrapping the module's own functions (the synchronisation serialises host and device:
the sum is an upper bound of the un-instrumented call, printed beside it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import init_device as idv, synth

dev = torch.device("cuda", 0)
n, nx, ny, k = 2048, 512, 512, int(os.environ.get("K", "5"))
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
Xd = synth.sample_torch(prob, dev, seed=1000)
if Xd.shape[0] != n:
    Xd = Xd.T     # (pixel-major sample: a view, as the estimator hands it)
Xh = np.empty((n, nx * ny), dtype=np.float32)   # (shape and dtype only)
acc = {}


def timed(name, fn):
    def w(*a, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **kw)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    W, H = idv.initialize_nmf_device(Xh, k, init="nndsvdar", random_state=0, X_device=Xd, X_mean=0.24)
    torch.cuda.synchronize()
    print(f"un-instrumented call {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)

orig = dict(lu=idv._lu_pl, qr=idv._qr_tall, nn=idv._nndsvd_long_on_device, mm=torch.Tensor.__matmul__, cpu=torch.Tensor.cpu)
idv._lu_pl = timed("LU normalisers (14)", orig["lu"])
idv._qr_tall = timed("Cholesky QR twice (incl. its products, read-backs)", orig["qr"])
idv._nndsvd_long_on_device = timed("NNDSVD post-processing on the device + H to the host", orig["nn"])


def mm(self, other):
    big = self.numel() >= n * nx * ny or other.numel() >= n * nx * ny
    if not big:
        return orig["mm"](self, other)
    return timed("products with X (16)", orig["mm"])(self, other)


torch.Tensor.__matmul__ = mm
for rep in range(2):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    W, H = idv.initialize_nmf_device(Xh, k, init="nndsvdar", random_state=0, X_device=Xd, X_mean=0.24)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"instrumented call {rep}: {1e3 * total:.2f} ms")
    for name, v in acc.items():
        print(f"   {name:62s} {1e3 * v:7.2f} ms")
    print(f"   {'everything else (random matrix, host SVD, small products, signs)':62s} {1e3 * (total - sum(acc.values())):7.2f} ms", flush=True)
