#!/usr/bin/env python3
"""Run-to-run reproducibility of the wide build's matrix-core kernels at the headline image (k = 16, 8-bit store): three
iterations twice from the same state, H and W compared bit for bit; then the iteration time.  ESPM_MU_WIDE_LIB selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

k, n, nx, ny = int(os.environ.get("K", "16")), 2048, 512, 512
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=3)
X = synth.sample_torch(prob, "cuda", seed=1003)
W0, H0 = synth.random_init(n, k, nx * ny, seed=3, scale=500.0 / n)
kw = dict(layout="pm", shape_2d=(nx, ny), lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False, tol=0.0, max_iter=140, x_store="u8", tile_px=256)
outs = []
for rep in range(3):
    eng = MUEngine(X, k, fused=True, **kw)
    eng.load_state(W0, H0)
    eng.iterate(3, final_loss=True)
    torch.cuda.synchronize()
    outs.append((eng.h[eng.st.cur][:, :eng.p].clone(), eng.w[eng.st.cur].clone(), eng.history()["loss"]))
for i in (1, 2):
    dh = (outs[i][0] != outs[0][0]).sum().item()
    dw = (outs[i][1] != outs[0][1]).sum().item()
    print(f"run {i} vs run 0: {dh} of {outs[0][0].numel()} entries of H differ (max {float((outs[i][0] - outs[0][0]).abs().max()):.3e}), {dw} of W; losses {outs[i][2][-1]:.9g} / {outs[0][2][-1]:.9g}")
eng.iterate(10, final_loss=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.iterate(100, final_loss=False)
torch.cuda.synchronize()
print(f"{os.environ.get('ESPM_MU_WIDE_LIB', 'product')}: k={k} {1e4 * (time.perf_counter() - t0):.1f} us / iteration")
