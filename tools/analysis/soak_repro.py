#!/usr/bin/env python3
"""Soak / reproducibility check of the C iteration loop (the W update's tail rides in the next H-step's launch): the same
fit twice at two doses, ITER iterations each; W, H and the whole history (loss, rel_W, rel_H) must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, nx, ny, k = 2048, 512, 512, 5
iters = int(os.environ.get("ITER", "1500"))
for dose in (18.0, 500.0):
    prob = synth.make_problem(n, nx, ny, k, N=dose, seed=0)
    X = synth.sample_torch(prob, dev, seed=1000)
    W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=dose / n)
    runs = []
    for rep in range(2):
        eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=iters,
                       device=dev)
        eng.load_state(W0, H0)
        done = 0
        while done < iters:          # chunks of uneven length: the flush at the end of a chunk and the restart are exercised too
            step = min(iters - done, 1 + (37 * (done + 1)) % 211)
            eng.iterate(step, final_loss=False)
            done += step
        eng.eval_current(advance_h=False)
        torch.cuda.synchronize()
        h = eng.history()
        runs.append((eng.get_W(), eng.get_H(), h))
        del eng
    (W1, H1, h1), (W2, H2, h2) = runs
    same = np.array_equal(W1, W2) and np.array_equal(H1, H2) and all(np.array_equal(h1[key], h2[key]) for key in h1)
    mono = bool((np.diff(h1["loss"]) <= 1e-9 * np.abs(h1["loss"][:-1])).all())
    print(f"N={dose:.0f}: {iters} iterations twice: bit-identical {same}; loss {h1['loss'][0]:.6f} -> {h1['loss'][-1]:.6f}, monotone {mono}; "
          f"nonfinite {h1['bad'].sum():.0f}; rel_W last {h1['rel_W'][-1]:.3e}, rel_W all finite {bool(np.isfinite(h1['rel_W'][1:]).all())}", flush=True)
    assert same
    del X
