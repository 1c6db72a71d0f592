#!/usr/bin/env python3
"""The headline image (2048 channels x 512 x 512 pixels, 500 counts per pixel) with 9..16 components: the second build of
the library (libespm_mu_wide.so, component stride 16) on the dense 8-bit store, next to k = 8 on the same store and on
the sparse one; 17..32 components: the third build (libespm_mu_wide32.so, stride 32).  K="8 12 16" chooses the component counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, nx, ny = 2048, 512, 512
for k in [int(v) for v in os.environ.get("K", "8 12 16").split()]:
    prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
    X = synth.sample_torch(prob, dev, seed=1000)
    W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=500.0 / n)
    # (wide builds: fused=False = the vector-ALU kernels, True = matrix cores; from 17 components on the 8-bit and bf16 stores have the matrix-core kernels only)
    for store, fused in ([("ell", True), ("u8", False), ("u8", True)] if k <= 16 else [("u8", True), ("bf16", True)]):
        eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=200,
                       device=dev, x_store=store, fused=fused) if not (store == "ell" and k > 16) else None
        if eng is None:
            continue
        eng.load_state(W0, H0)
        eng.iterate(10, final_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(100, final_loss=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        eng.eval_current(advance_h=False)
        h = eng.history()
        dense = 2 * n * nx * ny   # both copies of the 8-bit X are read once per iteration
        print(f"k={k:2d} store={eng.x_store:3s} {'mfma' if (fused and k >= 7 and store != 'ell') else 'valu'} (stride {eng.V.KP}): {dt * 1e6:7.1f} us/iteration = {1 / dt:6.0f} it/s"
              + (f"; X stream {dense * (2 if store == 'bf16' else 1) / dt / 1e12:.2f} TB/s" if store in ("u8", "bf16") else "")
              + f"; loss {h['loss'][0]:.6f} -> {h['loss'][-1]:.6f}; nonfinite {h['bad'].sum():.0f}", flush=True)
        del eng
    del X
