#!/usr/bin/env python3
"""Work balance of the sparse count store at the headline size: rows per list group, per H-step workgroup
(8 groups of one 512-pixel window) and per W-accumulation workgroup (the channel groups of a 1024-pixel block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

NX = NY = int(os.environ.get("NX", "512"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(2048, NX, NY, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
eng = MUEngine(X, 5, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=10, device=dev)
e = eng.ell
ho = e["ell_h_off"].cpu().numpy().astype(np.int64)
wo = e["ell_w_off"].cpu().numpy().astype(np.int64)
def stats(name, v):
    v = np.asarray(v, dtype=np.float64)
    print(f"{name:44s} n={v.size:6d} mean={v.mean():9.1f} min={v.min():9.1f} max={v.max():9.1f} max/mean={v.max()/v.mean():.3f}")
gh = ho[2::2] - ho[0:-1:2]
stats("H rows per group (one wave)", gh)
stats("H unit rows per group", ho[1::2] - ho[0:-1:2])
stats("H rows: longest wave of a workgroup", gh.reshape(-1, 8).max(axis=1))
stats("H rows: workgroup total", gh.reshape(-1, 8).sum(axis=1))
n_cg = e["n_cg"]
gw = (wo[2::2] - wo[0:-1:2]).reshape(e["nblk_w"], n_cg)
stats("W rows per (block, channel group)", gw.reshape(-1))
stats("W unit rows per (block, channel group)", (wo[1::2] - wo[0:-1:2]))
stats("W rows: block total", gw.sum(axis=1))
nw = 16
per_wave = np.zeros((e["nblk_w"], nw))
for t in range((n_cg + nw - 1) // nw):
    for w in range(nw):
        i = t * nw + ((nw - 1 - w) if (t & 1) else w)
        if i < n_cg:
            per_wave[:, w] += gw[:, i]
stats("W rows: longest wave of a block", per_wave.max(axis=1))
stats("W rows: mean wave of a block", per_wave.mean(axis=1))
print("entries_h", e["entries_h"], "rows_h*128", e["rows_h"] * 128, "entries_w", e["entries_w"], "rows_w*128", e["rows_w"] * 128, "nnz", e["nnz"])
