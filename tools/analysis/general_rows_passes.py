#!/usr/bin/env python3
"""LDS passes per 16-lane read group of the GENERAL rows' gathers at the headline size (tests/ell_decode.py: general_gather_passes), on a sample
of list groups, for the library ESPM_MU_LIB names (default: the product): what the builder's placement reaches on real list statistics.
ROWS (image rows, default 512), GROUPS (sample size, default 200)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from espm_amd import synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402
from ell_decode import general_gather_passes  # noqa: E402

ROWS, NG = int(os.environ.get("ROWS", "512")), int(os.environ.get("GROUPS", "200"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(2048, ROWS, 512, 5, N=500.0, seed=0, row0=0, nx_total=512)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
eng = MUEngine(X, 5, layout="pm", shape_2d=(ROWS, 512), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=10, device=dev)
del X
rng = np.random.default_rng(0)
for name, key, okey, bits in (("H lists", "ell_h", "ell_h_off", eng.st.ell_cbits), ("W lists", "ell_w", "ell_w_off", (2 * eng.st.tile_px).bit_length() - 1)):
    off = eng.ell[okey].cpu().numpy().astype(np.int64)
    ngroups = (len(off) - 1) // 2
    pick = np.sort(rng.choice(ngroups, size=min(NG, ngroups), replace=False))
    # a private copy of the sampled groups' rows with offsets of their own (the decoder walks every group it is given)
    words, noff, run = [], [0], 0
    full = eng.ell[key]
    for g in pick:
        w = full[off[2 * g] * 64:off[2 * g + 2] * 64].cpu().numpy()
        words.append(w)
        unit, gen = off[2 * g + 1] - off[2 * g], off[2 * g + 2] - off[2 * g + 1]
        noff += [run + unit, run + unit + gen]
        run += unit + gen
    passes, groups = general_gather_passes(np.concatenate(words), np.array(noff), bits)
    gen_rows = sum(noff[2 * i + 2] - noff[2 * i + 1] for i in range(len(pick)))
    print(f"{name}: {passes:.3f} passes per read group of the general rows ({groups} read groups, {gen_rows} general rows in {len(pick)} list groups; lib {os.environ.get('ESPM_MU_LIB', 'product')})")
