#!/usr/bin/env python3
"""Where a workgroup of the fused kernel spends its time at the headline size: phase stamps (100 MHz wall clock) written
by the instrumented build tools/analysis/libespm_mu_phase.so (build_phase_lib.sh).

stamps per workgroup: 0 entry | 1 GW table in LDS | 2 wave 0 walked its pixel lists | 3 every wave did | 4 wave 0 finished its
pixel's update | 5 records reduced, H' table complete | 6 wave 0 walked its channel lists | 7 every wave did
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ESPM_MU_LIB"] = os.path.join(ROOT, "tools", "analysis", os.environ.get("PHASE_LIB", "libespm_mu_phase.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from espm_amd import _lib, synth  # noqa: E402
from espm_amd.engine import MUEngine  # noqa: E402

C5 = os.environ.get("CONFIG") == "c5"   # CONFIG=c5 ROWS=128: a rank's share of BASELINE configuration 5 (1980 ch, 1024-pixel rows, k = 8, G 1980 x 17, mu = 0.05)
N_CH, NY, M = (1980, 1024, 17) if C5 else (2048, 512, None)
NX, K = int(os.environ.get("ROWS", "512")), 8 if C5 else int(os.environ.get("K", "5"))   # (ROWS < 512: the smaller block geometries)
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=float(os.environ.get("COUNTS", "500")), seed=0, row0=0, nx_total=NY, m=M)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(M if C5 else N_CH, K, NY * NY, seed=0, scale=500.0 / N_CH)
kw = dict(G=prob["G"], mu=0.05) if C5 else {}
eng = MUEngine(X, K, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=600, device=dev,
               fused={"0": False, "1": True}.get(os.environ.get("FUSED", "1"), os.environ.get("FUSED", "1")), **kw)
del X
eng.load_state(W0, H0[:, :NX * NY])
eng.iterate(300, final_loss=False)
torch.cuda.synchronize()
nblk = eng.st.nblk_w
SLOTS = 56
buf = torch.zeros((nblk + 1, SLOTS), dtype=torch.int64, device=dev)
fn = _lib.lib.espm_debug_phase_buffer
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
_lib.check(fn(C.c_void_p(buf.data_ptr())))
eng.iterate(3, final_loss=False)          # the stamps of the last of these stay
torch.cuda.synchronize()
t = buf[:nblk].cpu().numpy().astype(np.float64) * 0.01    # us
t0 = t[:, 0].min()
names = ["table -> LDS", "H walk (wave 0)", "H walk, slowest wave - wave 0", "per-pixel update", "record reduction + barrier", "W walk (wave 0)",
         "W walk, slowest wave - wave 0"]
print(f"{nblk} workgroups; kernel span (first entry -> last exit) {t[:, 7].max() - t0:7.2f} us; entries spread over {t[:, 0].max() - t0:5.2f} us")
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nm:36s} mean {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
tail = buf[nblk].cpu().numpy().astype(np.float64) * 0.01
if tail[7] > 0:
    print(f"  extra workgroup (tail of the previous W update): enters {tail[0] - t0:7.2f} us after the first entry, done at {tail[7] - t0:7.2f} us "
          f"(last regular workgroup exits at {t[:, 7].max() - t0:7.2f})")
tot = t[:, 7] - t[:, 0]
print(f"  {'workgroup total':36s} mean {tot.mean():7.2f}  min {tot.min():7.2f}  max {tot.max():7.2f} us")
print(f"  exits spread over {t[:, 7].max() - t[:, 7].min():5.2f} us")
if t[:, 40].max() > 0:   # stamps inside the per-pixel update (thread 0): partial sums | regularisers + stencil (loads arrived) | simplex root | stores (= stamp 4)
    sub = [("sum of the partial numerators", 3, 40), ("regularisers, stencil (loads arrived)", 40, 41), ("simplex multiplier", 41, 42), ("H', table row, statistics", 42, 4)]
    for nm, a_, b_ in sub:
        d = t[:, b_] - t[:, a_]
        print(f"    {nm:38s} mean {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
# per wave: end of the H walk / of the W walk relative to the stamp that started it (1: table ready, 5: H' table ready)
hw = t[:, 8:24] - t[:, 1:2]
ww = t[:, 24:40] - t[:, 5:6]
print("H walk per wave (mean over workgroups, us): " + " ".join(f"{v:5.1f}" for v in hw.mean(axis=0)))
print("W walk per wave (mean over workgroups, us): " + " ".join(f"{v:5.1f}" for v in ww.mean(axis=0)))
ell = eng.ell
off = ell["ell_h_off"].cpu().numpy()
rows = off[2::2] - off[0:-1:2]
unit = off[1::2] - off[0:-1:2]
gpt = max(eng.st.tile_px // 64, 1)
g = rows.reshape(-1, gpt)
u = unit.reshape(-1, gpt)
print("H list groups of a window, rows (mean over windows):        " + " ".join(f"{v:6.1f}" for v in g.mean(axis=0)))
print("                                   of which unit rows:                " + " ".join(f"{v:6.1f}" for v in u.mean(axis=0)))
