#!/usr/bin/env python3
"""Fused half-steps (one launch for H update + W accumulation, include/espm_mu.h: no_fused) against the two launches at
the headline size: same W, H, losses after 6 iterations, and the iteration time of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from espm_amd import synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402

N_CH, NX, NY, K = 2048, 512, 512, int(os.environ.get("K", "5"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
W0, H0 = synth.random_init(N_CH, K, NX * NY, seed=0, scale=500.0 / N_CH)
kw = dict(layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=1300, device=dev)
out = {}
engs = {}
for fused in (False, "static", True):
    eng = MUEngine(X, K, fused=fused, **kw)
    eng.load_state(W0, H0)
    eng.iterate(6, final_loss=True)
    torch.cuda.synchronize()
    out[fused] = (eng.get_W(), eng.get_H(), eng.history())
    if fused:   # bit-reproducible run to run although the waves take their work from a counter
        eng.load_state(W0, H0)
        eng.iterate(6, final_loss=True)
        torch.cuda.synchronize()
        h2 = eng.history()
        assert np.array_equal(eng.get_W(), out[fused][0]) and np.array_equal(eng.get_H(), out[fused][1])
        assert np.array_equal(h2["loss"], out[fused][2]["loss"]), (h2["loss"], out[fused][2]["loss"])
        print("fused: bitwise reproducible")
    engs[fused] = eng
for rep in range(4):      # interleaved A/B on one device: the clocks and the neighbours are the same for all variants
    for fused, eng in engs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(300, final_loss=False)
        torch.cuda.synchronize()
        print(f"fused={fused!s:7s}: {(time.perf_counter() - t0) / 300 * 1e6:7.1f} us / iteration", flush=True)
(Wa, Ha, ha), (Wb, Hb, hb) = out[False], out[True]
print("max |dW| / max W", np.abs(Wa - Wb).max() / Wa.max(), " max |dH|", np.abs(Ha - Hb).max())
print("losses two launches:", ha["loss"])
print("losses fused       :", hb["loss"])
print("rel_W", ha["rel_W"], hb["rel_W"])
print("rel_H", ha["rel_H"], hb["rel_H"])
np.testing.assert_allclose(Wa, Wb, rtol=2e-5, atol=1e-9)
np.testing.assert_allclose(Ha, Hb, rtol=2e-5, atol=1e-7)
np.testing.assert_allclose(ha["loss"], hb["loss"], rtol=2e-7)   # (the fused walk sums a pixel's numerator in up to four parts)
print("FUSED_OK")
