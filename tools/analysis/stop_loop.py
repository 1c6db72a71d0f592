#!/usr/bin/env python3
"""Cost of the host-checked stop criterion (the estimator's default) per iteration at the headline size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, contextlib, io
from espm_amd import synth
from espm_amd.estimators import SmoothNMF
prob = synth.make_problem(2048, 512, 512, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1000).t().contiguous().cpu().numpy()
W0, H0 = synth.random_init(2048, 5, 512 * 512, seed=0, scale=500.0 / 2048)
for label, kw in (("no stop criterion", dict(tol=0, no_stop_criterion=True)), ("stop criterion checked every iteration", dict(tol=1e-30))):
    est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(512, 512), max_iter=300, verbose=0, **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        est.fit_transform(X, W=W0.copy(), H=H0.copy())   # warm
        t0 = time.perf_counter(); est.fit_transform(X, W=W0.copy(), H=H0.copy()); dt = time.perf_counter() - t0
    print(f"{label}: {dt:.3f} s for {est.n_iter_} iterations")
