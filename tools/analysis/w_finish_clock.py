#!/usr/bin/env python3
"""Where the one-workgroup W finish (simplex over W: the reference's default constraint) spends its time at the headline
size: phase stamps (100 MHz wall clock) of the instrumented build tools/analysis/libespm_mu_phase.so (build_phase_lib.sh).

stamps: 0 entry | 1 G^T A (dictionary G only) | 2 numerators, denominators, bracket | 3 roots (Newton) | 4 the sweep the
reference's bisection stops at | 5 W', rel_W | 6 rows of G W', column sums; slot 8: evaluations of f by the root finder"""
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

N_CH, NX, NY, K = 2048, 512, 512, int(os.environ.get("K", "5"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)
W0, H0 = synth.random_init(N_CH, K, NX * NY, seed=0, scale=500.0 / N_CH)
eng = MUEngine(X, K, layout="pm", shape_2d=(NX, NY), lambda_L=0.0, simplex_H=False, simplex_W=True, tol=0.0, max_iter=200, device=dev)
del X
eng.load_state(W0 / W0.sum(axis=0, keepdims=True), H0)
eng.iterate(60, final_loss=False)
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
d = (t[:, 1:7] - t[:, 0:6]) * 0.01
names = ["G^T A", "num / den / bracket", "roots (Newton)", "stopping sweep", "W', rel_W", "G W' rows, column sums"]
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f} us")
fine = (t[:, [10, 2]] - t[:, [1, 10]]) * 0.01
print("  inside 'num / den / bracket': loads + num / den %.2f | six reductions over the 16 waves + owner set-up %.2f us" % tuple(fine.mean(axis=0)))
print(f"  shader clock during the kernel: {((t[:, 21] - t[:, 20]) / ((t[:, 6] - t[:, 0]) * 0.01)).mean():.0f} ticks per us")
print(f"  total {((t[:, 6] - t[:, 0]) * 0.01).mean():6.2f} us; evaluations of f in the root finder: {t[:, 8].mean():.1f}")
