#!/usr/bin/env python3
"""Which workgroups of the fused kernel are the slow ones - the same CUs every launch (hardware), the same blocks (data), or neither?

Instrumented build (build_phase_lib.sh): per workgroup the phase stamps and where it ran (XCC, SE, CU from HW_ID / XCC_ID).
REPS launches from consecutive states; per launch the workgroup's total, H-walk and W-walk time.  Printed: the spread, the
correlation of a workgroup's time between launches (by block = by CU if the placement repeats), the means by XCC and by CU slot,
and whether block -> CU placement repeats from launch to launch."""
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

N_CH, NX, NY, K = 2048, int(os.environ.get("ROWS", "512")), 512, int(os.environ.get("K", "5"))
REPS = int(os.environ.get("REPS", "12"))
dev = torch.device("cuda", 0)
prob = synth.make_problem(N_CH, NX, NY, K, N=500.0, seed=0, row0=0, nx_total=512)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(N_CH, K, 512 * 512, seed=0, scale=500.0 / N_CH)
eng = MUEngine(X, K, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=900, device=dev)
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
tot, hw_, ww, where, entry = [], [], [], [], []
for rep in range(REPS):
    eng.iterate(2, final_loss=False)
    torch.cuda.synchronize()
    raw = buf[:nblk].cpu().numpy()
    t = raw.astype(np.float64) * 0.01
    tot.append(t[:, 7] - t[:, 0])
    hw_.append(t[:, 3] - t[:, 1])
    ww.append(t[:, 7] - t[:, 5])
    entry.append(t[:, 0] - t[:, 0].min())
    w = raw[:, 43]
    hwid, xcc = w & 0xffffffff, (w >> 32) & 0xf
    cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7
    where.append(np.stack([xcc, se, sh, cu], axis=1))
tot, hw_, ww, where, entry = map(np.array, (tot, hw_, ww, where, entry))
print(f"{nblk} workgroups x {REPS} launches; workgroup total: mean {tot.mean():.2f} us, per-launch max-mean {np.mean(tot.max(axis=1) - tot.mean(axis=1)):.2f}, "
      f"max-min {np.mean(tot.max(axis=1) - tot.min(axis=1)):.2f}")
same = [(where[i] == where[0]).all(axis=1).mean() for i in range(1, REPS)]
print("block -> (xcc, se, sh, cu) placement equal to launch 0: " + " ".join(f"{v:.2f}" for v in same))
print("distinct (xcc, se, sh, cu) in launch 0:", len({tuple(r) for r in where[0]}), " XCCs:", sorted(set(where[0][:, 0])),
      " blocks per XCC:", np.bincount(where[0][:, 0]).tolist())
c = np.corrcoef(tot)
print(f"correlation of a BLOCK's total time between launches: mean off-diagonal {((c.sum() - REPS) / (REPS * (REPS - 1))):.3f}")
for name, arr in (("total", tot), ("H walk", hw_), ("W walk", ww)):
    m = arr.mean(axis=0)
    print(f"  {name:7s}: per-block mean over launches: min {m.min():.2f} max {m.max():.2f} std {m.std():.2f};  within-block std over launches {arr.std(axis=0).mean():.2f}")
# by hardware unit (keyed over all launches)
keys = {}
for i in range(REPS):
    for b in range(nblk):
        keys.setdefault(tuple(where[i, b]), []).append(tot[i, b])
unit_mean = {k: np.mean(v) for k, v in keys.items()}
vals = np.array(list(unit_mean.values()))
print(f"by hardware unit ({len(vals)} units): mean of unit means {vals.mean():.2f}, min {vals.min():.2f}, max {vals.max():.2f}, std {vals.std():.2f}")
for x in sorted(set(where[..., 0].ravel())):
    sel = where[..., 0] == x
    print(f"  XCC {x}: total {tot[sel].mean():.2f}  H walk {hw_[sel].mean():.2f}  W walk {ww[sel].mean():.2f}  entry +{entry[sel].mean():.2f} us")
# the slowest blocks of launch 0: are they slow again?
order = np.argsort(-tot.mean(axis=0))[:8]
print("slowest blocks (mean over launches): " + ", ".join(f"b{b}: {tot[:, b].mean():.1f} (std {tot[:, b].std():.1f}) @ {tuple(where[0, b])}" for b in order))
rows = eng.ell["ell_h_off"].cpu().numpy()
print("entries of those blocks vs mean: pixel-list rows", [(int(rows[2 * 16 * (b + 1)] - rows[2 * 16 * b])) for b in order], "mean", float((rows[-1] - rows[0]) / nblk))
