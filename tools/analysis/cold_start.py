#!/usr/bin/env python3
"""Why the first tens of iterations after the set-up run slower (the driver times 20 steps after 5 warm-up iterations):
per-iteration device time by HIP events right after the engine is built, after a busy phase, and after an idle second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

N_CH, NX, NY, K = 2048, 512, 512, 5
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=500.0, seed=0)
W0, H0 = synth.random_init(N_CH, K, NX * NY, seed=0, scale=500.0 / N_CH)
X = synth.sample_torch(prob, dev, seed=1000)
eng = MUEngine(X, K, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=3000, device=dev)
W0d, H0d = torch.from_numpy(W0).to(dev, torch.float32), torch.from_numpy(H0).to(dev, torch.float32)
eng.load_state(W0d, H0d)


def per_iteration(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        eng.iterate(1, final_loss=False)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)])


def show(tag, t):
    print(f"{tag:34s} it 1-5 {t[:5].mean():6.1f} | 6-25 {t[5:25].mean():6.1f} | 26-60 {t[25:60].mean():6.1f} | 61-120 {t[60:120].mean():6.1f} | 121-300 {t[120:300].mean():6.1f} us", flush=True)


show("right after the set-up", per_iteration(300))
del eng
X2 = synth.sample_torch(prob, dev, seed=1000)
eng = MUEngine(X2, K, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=3000, device=dev)
eng.load_state(W0d, H0d)
show("second engine, right after set-up", per_iteration(300))
show("straight on (busy for 45 ms)", per_iteration(300))
time.sleep(1.0)
show("after 1 s idle", per_iteration(300))
time.sleep(0.05)
show("after 50 ms idle", per_iteration(300))
time.sleep(0.005)
show("after 5 ms idle", per_iteration(300))
try:
    import subprocess
    print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
