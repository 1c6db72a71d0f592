#!/usr/bin/env python3
"""The headline image with the reference's DEFAULT constraints (simplex_W=True, simplex_H=False: the W update then needs a
multiplier per component over all channels - one workgroup finishes W) next to the headline's simplex_H."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, nx, ny, k = 2048, 512, 512, int(os.environ.get("K", "5"))
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=500.0 / n)
W0 = W0 / W0.sum(axis=0, keepdims=True)
for name, kw in (("simplex_H + Laplacian (headline)", dict(lambda_L=1.0, simplex_H=True, simplex_W=False)),
                 ("simplex_W (reference default)", dict(lambda_L=0.0, simplex_H=False, simplex_W=True)),
                 ("simplex_W + Laplacian + mu", dict(lambda_L=1.0, mu=0.05, simplex_H=False, simplex_W=True)),
                 ("no simplex", dict(lambda_L=0.0, simplex_H=False, simplex_W=False))):
    eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), tol=0.0, max_iter=200, device=dev, **kw)
    eng.load_state(W0, H0 * (500.0 if kw["simplex_W"] else 1.0))
    eng.iterate(10, final_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(100, final_loss=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    eng.eval_current(advance_h=False)
    h = eng.history()
    print(f"{name:34s}: {dt * 1e6:7.1f} us/iteration = {1 / dt:6.0f} it/s; loss {h['loss'][0]:.5f} -> {h['loss'][-1]:.5f}; nonfinite {h['bad'].sum():.0f}", flush=True)
    del eng
