#!/usr/bin/env python3
"""Is a fitted estimator freed when its last reference goes (reference counting), or only by the cyclic collector - which then
unmaps its 2 GB X_ at a random moment of a later fit?  Fits a 512 x 96 x 96 image on the device path and looks."""
import gc, os, sys, weakref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.estimators import SmoothNMF

gc.disable()
prob = synth.make_problem(512, 96, 96, 3, N=200.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1).t().contiguous().cpu().numpy().astype(np.float32)
est = SmoothNMF(n_components=3, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(96, 96), max_iter=20, tol=0, no_stop_criterion=True, verbose=0,
                random_state=0)
est.fit_transform(X)
print("X_ is", type(est.X_).__name__, "of", est.X_.shape)
wr_est, wr_x = weakref.ref(est), weakref.ref(est.X_)
eng = getattr(est, "_engine", None)
wr_eng = weakref.ref(eng) if eng is not None else None
del eng
del est
print("after del, before gc.collect(): estimator alive:", wr_est() is not None, " X_ alive:", wr_x() is not None,
      " engine alive:", (wr_eng() is not None) if wr_eng else None)
if wr_est() is not None or wr_x() is not None or (wr_eng and wr_eng() is not None):
    for name, wr in (("estimator", wr_est), ("X_", wr_x), ("engine", wr_eng)):
        o = wr() if wr else None
        if o is None:
            continue
        print(f"referrers of the {name}:")
        for r in gc.get_referrers(o):
            if r is locals() or r is globals():
                continue
            d = repr(type(r))
            extra = ""
            if isinstance(r, dict):
                extra = " keys: " + ", ".join(list(map(str, r.keys()))[:12])
            elif hasattr(r, "__qualname__"):
                extra = " " + r.__qualname__
            elif hasattr(r, "f_code"):
                extra = f" frame {r.f_code.co_name}"
            print("    ", d, extra[:200])
        del o
n = gc.collect()
print("gc.collect() freed", n, "objects; estimator alive:", wr_est() is not None, " X_ alive:", wr_x() is not None)
