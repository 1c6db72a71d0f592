#!/usr/bin/env python3
"""A/B of builds of the narrow library in ONE process, on one device, interleaved: same image, same state.

    python tools/analysis/variant_ab.py base=espm_amd/lib/libespm_mu.so klprod=tools/analysis/libespm_mu_klprod.so ...

env: ROWS (image rows of the 512-wide headline image: 512 = the headline, 64 = an eighth), K, COUNTS, FUSED (engine's `fused`),
ITERS (per timing slice), REPS.  Every build runs 6 iterations from the same state first: losses, W, H against the first build."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from espm_amd import _lib, synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402

ROWS, K = int(os.environ.get("ROWS", "512")), int(os.environ.get("K", "5"))
COUNTS = float(os.environ.get("COUNTS", "500"))
ITERS, REPS = int(os.environ.get("ITERS", "300")), int(os.environ.get("REPS", "4"))
fused = {"0": False, "1": True}.get(os.environ.get("FUSED", "1"), os.environ.get("FUSED", "1"))
dev = torch.device("cuda", 0)
C5 = os.environ.get("CONFIG") == "c5"   # CONFIG=c5 ROWS=128: a rank's share of BASELINE configuration 5 (1980 ch, 1024-pixel rows, k = 8, G 1980 x 17, mu = 0.05)
N_CH, NY, M = (1980, 1024, 17) if C5 else (2048, 512, None)
if C5:
    K = 8
prob = synth.make_problem(N_CH, ROWS, NY, K, N=COUNTS, seed=0, row0=0, nx_total=NY, m=M)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(M if C5 else N_CH, K, NY * NY, seed=0, scale=COUNTS / N_CH)
H0 = H0[:, :ROWS * NY]
kw = dict(layout="pm", shape_2d=(ROWS, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=ITERS * (REPS + 1) + 20, device=dev, fused=fused)
if C5:
    kw.update(G=prob["G"], mu=0.05)
engs, ref = {}, None
ENVS = {}   # name -> environment the build is RUN under (name=path@VAR=value,VAR2=value: knobs the library reads per launch)
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    if "@" in path:
        path, envs = path.split("@", 1)
        ENVS[name] = dict(e.split("=", 1) for e in envs.split(","))
    os.environ.update(ENVS.get(name, {}))
    handle = _lib._load(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    _lib._narrow = _lib.Variant(handle, _lib.KP, 1, _lib.MAX_K)
    eng = MUEngine(X, K, **kw)
    eng.load_state(W0, H0)
    eng.iterate(6, final_loss=True)
    torch.cuda.synchronize()
    out = (eng.get_W(), eng.get_H(), eng.history()["loss"])
    if ref is None:
        ref = out
        print(f"{name:10s}: losses {out[2]}")
    else:
        dl = np.max(np.abs(out[2] - ref[2]) / np.abs(ref[2]))
        print(f"{name:10s}: max rel dloss {dl:.2e}  max|dW|/max W {np.abs(out[0] - ref[0]).max() / ref[0].max():.2e}  max|dH| {np.abs(out[1] - ref[1]).max():.2e}")
        assert dl < 2e-6 or os.environ.get("AB_NO_CHECK") == "1"   # (AB_NO_CHECK=1: a timing-only experiment that computes something else)
    eng.load_state(W0, H0)
    eng.iterate(6, final_loss=True)
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_W(), out[0]) and np.array_equal(eng.history()["loss"], out[2]), f"{name}: not reproducible run to run"
    engs[name] = eng
    print(f"{name:10s}: store {eng.x_store}, {eng.x_bytes / 2 ** 20:.0f} MB, ell_pb {eng.st.ell_pb}, ell_stream {eng.st.ell_stream}", flush=True)
    for var in ENVS.get(name, {}):
        os.environ.pop(var, None)
del X
best = {n: 1e9 for n in engs}
for rep in range(REPS + 1):      # (the first round is the run-in)
    for name, eng in engs.items():
        os.environ.update(ENVS.get(name, {}))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(ITERS, final_loss=False)
        torch.cuda.synchronize()
        for var in ENVS.get(name, {}):
            os.environ.pop(var, None)
        us = (time.perf_counter() - t0) / ITERS * 1e6
        if rep:
            best[name] = min(best[name], us)
        print(f"rep {rep} {name:10s}: {us:7.1f} us / iteration", flush=True)
print("best: " + "  ".join(f"{n} {v:.1f}" for n, v in best.items()))
