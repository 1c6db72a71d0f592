#!/usr/bin/env python3
"""A serialized timeline of SmoothNMF.fit_transform at the headline size: the phases of the fit wrapped in device
synchronisations (so each is charged its own device work; the real fit overlaps some of it and is a little shorter)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.estimators import SmoothNMF, base
from espm_amd.engine import MUEngine

marks = []


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        marks.append((name, t0 - T0[0], time.perf_counter() - t0))
        return out
    return wrapper


T0 = [0.0]
base.initialize_algorithms = timed("initialize_algorithms (NNDSVD on the device)", base.initialize_algorithms)
base.NMFEstimator._make_engine = timed("engine set-up (pack, sparse store build, state)", base.NMFEstimator._make_engine)
MUEngine.iterate = timed("iterate", MUEngine.iterate)
MUEngine.history = timed("history read-back", MUEngine.history)
MUEngine.get_W = timed("get_W", MUEngine.get_W)
if os.environ.get("NOCOPY") == "1":   # experiment: the fit without the estimator's host copy X_ (is the copy what unsettles the phases?)
    base._HostCopy.start = lambda self: None
    base._HostCopy.result = lambda self: None
else:
    base._HostCopy.result = timed("join of the host copy X_", base._HostCopy.result)
_to = torch.Tensor.to


def to(self, *a, **k):
    big = self.numel() > 1 << 24 and not self.is_cuda
    if not big:
        return _to(self, *a, **k)
    return timed("upload of X", _to)(self, *a, **k)


torch.Tensor.to = to
prob = synth.make_problem(2048, 512, 512, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1000).t().contiguous().cpu().numpy().astype(np.float32)
keep = None
if os.environ.get("KEEPWARM") == "1":   # experiment: a trickle of small kernels on a side stream for the whole run (do the phases steady?)
    import threading
    stop_warm = threading.Event()

    def warm():
        side = torch.cuda.Stream()
        a = torch.rand(256, 256, device="cuda")
        with torch.cuda.stream(side):
            while not stop_warm.is_set():
                for _ in range(50):
                    a = (a @ a).clamp_(0, 1)
                side.synchronize()
    keep = threading.Thread(target=warm, daemon=True)
    keep.start()
for rep in range(4):
    marks.clear()
    est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(512, 512), max_iter=200, tol=0,
                    no_stop_criterion=True, verbose=0, random_state=0)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    T0[0] = time.perf_counter()
    est.fit_transform(X)
    torch.cuda.synchronize()
    total = time.perf_counter() - T0[0]
    if rep:
        print(f"rep {rep}: fit_transform {total:.3f} s (serialized)")
        for name, at, dt in marks:
            print(f"   at {at * 1e3:7.1f} ms  {dt * 1e3:7.1f} ms  {name}")
        print(f"   unaccounted (host code between the phases): {(total - sum(m[2] for m in marks)) * 1e3:7.1f} ms")
        ms1 = torch.cuda.memory_stats()
        print("   device allocator during the fit: %d hipMalloc, %d hipFree, peak reserved %.1f GB" % (
            ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"],
            ms1["reserved_bytes.all.peak"] / 1e9))
if keep is not None:
    stop_warm.set()
    keep.join()
