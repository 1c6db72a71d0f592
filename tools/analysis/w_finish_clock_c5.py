#!/usr/bin/env python3
"""Phases of the one-workgroup W finish with a dictionary G (BASELINE config 5's shard: 1980 channels, G 1980 x 17, k = 8):
the stamps of tools/analysis/w_finish_clock.py on the instrumented build."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ESPM_MU_LIB"] = os.path.join(ROOT, "tools", "analysis", "libespm_mu_phase.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from espm_amd import _lib, synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402

ROWS = int(os.environ.get("ROWS", "128"))
n, ny, k, m = 1980, 1024, 8, 17
dev = torch.device("cuda", 0)
prob = synth.make_problem(n, ROWS, ny, k, N=500.0, seed=0, m=m, row0=0, nx_total=1024)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(m, k, 1024 * ny, seed=0, scale=0.3)
eng = MUEngine(X, k, G=prob["G"], layout="pm", shape_2d=(ROWS, ny), lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False,
               tol=0.0, max_iter=200, device=dev)
del X
eng.load_state(W0, H0[:, :ROWS * ny])
eng.iterate(40, final_loss=False)
torch.cuda.synchronize()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
fn = _lib.lib.espm_debug_phase_buffer_w
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
_lib.check(fn(C.c_void_p(buf.data_ptr())))
rows = []
for _ in range(8):
    eng.iterate(1, final_loss=False)
    torch.cuda.synchronize()
    rows.append(buf.cpu().numpy().astype(np.float64))
t = np.array(rows)
names = {1: "G^T A", 10: "loads, numerators, denominators", 4: "(no simplex)", 5: "W', rel_W", 6: "rows of G W', column sums"}
order = [0, 1, 10, 4, 5, 6]
for a, b in zip(order[:-1], order[1:]):
    d = (t[:, b] - t[:, a]) * 0.01
    print(f"  {names[b]:34s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
print(f"  total {((t[:, 6] - t[:, 0]) * 0.01).mean():6.2f} us")
