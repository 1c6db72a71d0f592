#!/usr/bin/env python3
"""The headline image (2048 channels x 512 x 512 pixels, k = 5, simplex_H + Laplacian) at other doses: N counts per
pixel from DOSES (default "5 18 100 500 2000"), stores from STORES (default "auto").  Low doses leave pixels without
counts (their fill's numerator comes from the small extra pass); high doses leave the sparse store
for the dense 8-bit one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, nx, ny, k = 2048, 512, 512, 5
for dose in [float(v) for v in os.environ.get("DOSES", "5 18 100 500 2000").split()]:
    prob = synth.make_problem(n, nx, ny, k, N=dose, seed=0)
    X = synth.sample_torch(prob, dev, seed=1000)
    W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=dose / n)
    nnz = float((X != 0).sum()) / X.numel()
    for store in os.environ.get("STORES", "auto").split():
        t0 = time.perf_counter()
        eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=200,
                       device=dev, x_store=store)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        eng.load_state(W0, H0)
        eng.iterate(10, final_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(100, final_loss=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        eng.eval_current(advance_h=False)
        h = eng.history()
        print(f"N={dose:6.0f} counts/pixel: {100 * nnz:5.2f} % non-zero, store={eng.x_store:3s}, pixels without counts {int(eng.st.ell_fill_n):6d}, "
              f"engine set-up {t_build * 1e3:6.1f} ms, {dt * 1e6:7.1f} us/iteration = {1 / dt:6.0f} it/s; loss {h['loss'][0]:.6f} -> {h['loss'][-1]:.6f}; "
              f"nonfinite {h['bad'].sum():.0f}", flush=True)
        del eng
    del X
