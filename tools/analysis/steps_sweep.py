#!/usr/bin/env python3
"""How the measured iteration rate depends on the length of the timed region: the driver times 20 steps after 5
warm-up iterations, the default bench 300 after 30.  One engine, the timing bracket of bench.py, several lengths."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from espm_amd import synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402

N_CH, NX, NY, K = 2048, 512, 512, 5
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
W0, H0 = synth.random_init(N_CH, K, NX * NY, seed=0, scale=500.0 / N_CH)
eng = MUEngine(X, K, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4000, device=dev)
del X
eng.load_state(W0, H0)
torch.cuda.synchronize()


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(n, final_loss=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("first 5 (warm-up) us/it:", round(timed(5), 1))
for n in (20, 20, 20, 20, 100, 300, 20, 20, 300):
    print(f"steps {n:4d}: {timed(n):7.1f} us/it", flush=True)
time.sleep(2.0)
print("after 2 s idle: 5 warm-up", round(timed(5), 1), " then 20:", round(timed(20), 1), flush=True)
