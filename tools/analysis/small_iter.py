#!/usr/bin/env python3
"""Iterations of the C loop on images that do not fill the chip with 1024-pixel blocks: ROWS image rows of the headline
problem (2048 channels, 512 pixels per row, k = 5; ROWS = 64: what one rank of an 8-GPU run holds).  FUSED=0: the two-kernel
path (A/B).  For rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

ROWS = [int(v) for v in os.environ.get("ROWS", "64").split(",")]
FUSED = os.environ.get("FUSED", "1") != "0"
K = int(os.environ.get("K", "5"))   # (components: the segments per list group of the fused kernel follow it)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rows in ROWS:
    prob = synth.make_problem(2048, rows, 512, K, N=500.0, seed=0, row0=0, nx_total=512)
    X = synth.sample_torch(prob, dev, seed=1000, row0=0)
    W0, H0 = synth.random_init(2048, K, 512 * 512, seed=0, scale=500.0 / 2048)
    eng = MUEngine(X, K, layout="pm", shape_2d=(rows, 512), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=2000,
                   device=dev, fused=FUSED)
    eng.load_state(W0, H0[:, :rows * 512])
    eng.iterate(100, final_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(1000, final_loss=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 1000
    print(f"k {K} rows {rows}: tile_px {eng.st.tile_px} ell_pb {eng.st.ell_pb} nblk_w {eng.st.nblk_w} fused {FUSED}: {dt * 1e6:.1f} us/iteration", flush=True)
    del eng
