#!/usr/bin/env python3
"""The reference's default constraints (simplex over W, G = identity) at the headline size: the W update as two many-workgroup
launches against the one-workgroup finish (fused=False also keeps the H-step and the W accumulation as two launches).
Same W to rounding, per-iteration times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

n, nx, ny, k = 2048, 512, 512, int(os.environ.get("K", "5"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=500.0 / n)
W0 /= W0.sum(axis=0, keepdims=True)
res = {}
for name, fused in (("split", True), ("one workgroup", False)):
    eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=0.0, simplex_H=False, simplex_W=True, tol=0.0, max_iter=400, device=dev, fused=fused)
    eng.load_state(W0, H0)
    eng.iterate(50, final_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(300, final_loss=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    res[name] = (eng.get_W().astype(np.float64), eng.history()["loss"])
    print(f"{name:14s}: {dt * 1e6:6.1f} us/iteration; loss {res[name][1][-1]:.9f}; column sums of W {res[name][0].sum(axis=0)}", flush=True)
    del eng
a, b = res["split"], res["one workgroup"]
print("max |W_split - W_one| / max W = %.2e; max rel loss difference over 350 iterations %.2e" %
      (np.abs(a[0] - b[0]).max() / b[0].max(), np.abs(a[1] / b[1] - 1).max()))
