#!/usr/bin/env python3
"""Five fits of the headline problem (200 iterations, host fp32 array in, host arrays out) with the estimator's own host-side
time stamps (ESPM_FIT_TIMING=1: no device synchronisation added): which section of fit_transform the time sits in."""
import os, sys, time
os.environ["ESPM_FIT_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, io
import numpy as np
import torch
from espm_amd import synth
from espm_amd.estimators import SmoothNMF

prob = synth.make_problem(2048, 512, 512, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1000).t().contiguous().cpu().numpy().astype(np.float32)
for rep in range(6):
    est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(512, 512), max_iter=200, tol=0,
                    no_stop_criterion=True, verbose=0, random_state=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        est.fit_transform(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        print(f"rep {rep}: fit_transform {dt:.3f} s")
        print("   " + "\n   ".join(l for l in buf.getvalue().splitlines() if l.startswith("[fit timing")))
