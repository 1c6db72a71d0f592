#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: 1980 channels x (1024 x 1024) pixels, k = 8, fixed dictionary G (1980 x 17),
mu = 0.05, lambda = 1, simplex_H; and its 128-row shard (what one of 8 ranks owns).  ROWS=128 for the shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

ROWS = int(os.environ.get("ROWS", "1024"))
STORE = os.environ.get("STORE", "auto")
ITERS = int(os.environ.get("ITERS", "200"))   # (a --pmc pass serialises the launches: ITERS=30 there)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, ny, k, m = 1980, 1024, 8, 17
prob = synth.make_problem(n, ROWS, ny, k, N=500.0, seed=0, m=m, row0=0, nx_total=1024)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(m, k, 1024 * ny, seed=0, scale=0.3)
eng = MUEngine(X, k, G=prob["G"], layout="pm", shape_2d=(ROWS, ny), lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False,
               tol=0.0, max_iter=400, device=dev, x_store=STORE)
del X
eng.load_state(W0, H0[:, :ROWS * ny])
eng.iterate(20, final_loss=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.iterate(ITERS, final_loss=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / ITERS
eng.eval_current(advance_h=False)
h = eng.history()
print(f"C5 rows={ROWS} store={eng.x_store}: {dt * 1e6:.1f} us/iteration = {1 / dt:.0f} it/s; loss {h['loss'][0]:.6f} -> {h['loss'][-1]:.6f}; nonfinite {h['bad'].sum():.0f}")
