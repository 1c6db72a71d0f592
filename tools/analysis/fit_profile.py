#!/usr/bin/env python3
"""Where the wall time of a whole SmoothNMF.fit_transform goes at the headline size (host array in, host arrays out)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.estimators import SmoothNMF

dt = np.float32 if os.environ.get("DTYPE", "f32") == "f32" else np.float64
prob = synth.make_problem(2048, 512, 512, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1000).t().contiguous().cpu().numpy().astype(dt)   # (n, p) host
print("X", X.shape, X.dtype, "%.2f GB" % (X.nbytes / 1e9))
for rep in range(2):
    est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(512, 512), max_iter=200, tol=0,
                    no_stop_criterion=True, verbose=0, random_state=0)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    est.fit_transform(X)
    pr.disable()
    torch.cuda.synchronize()
    print(f"rep {rep}: fit_transform {time.perf_counter() - t0:.3f} s, n_iter {est.n_iter_}, loss {est.losses_[-1]:.6f}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).print_callers("sum|reduce|method .to.|method .cpu.")
print(s.getvalue()[:5000])
