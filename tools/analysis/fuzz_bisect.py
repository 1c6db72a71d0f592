#!/usr/bin/env python3
"""Which ingredient of a failing fuzz draw breaks: SEED=15 python tools/analysis/fuzz_bisect.py (the fused kernel forced)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ESPM_FORCE_ELL_TILE"] = os.environ.get("TILE", "512")
import numpy as np
from test_gpu_fuzz import _case
from oracle import mu_oracle as oc
from espm_amd.estimators import SmoothNMF

seed = int(os.environ.get("SEED", "15"))
base = _case(seed)
print(base["algo"], base["k"], base["kw"], sorted(base["extra"]), base["shape"], base["X"].shape)
variants = {"as drawn": {}, "no linesearch": {"drop_extra": ["linesearch"]}, "no fixed_H": {"drop_extra": ["fixed_H"]},
            "no simplex_W": {"kw": {"simplex_W": False}}, "no mu": {"kw": {"mu": 0}}, "no lambda": {"kw": {"lambda_L": 0.0}, "drop_extra": ["linesearch"]}}
for name, v in variants.items():
    c = dict(base); kw = dict(c["kw"]); kw.update(v.get("kw", {})); extra = {k: e for k, e in c["extra"].items() if k not in v.get("drop_extra", [])}
    ref = oc.fit(c["X"], c["k"], G=c["G"], W=c["W0"].copy(), H=c["H0"].copy(), shape_2d=c["shape"], algo=c["algo"], tol=0, no_stop_criterion=True,
                 max_iter=4, exact_root=(c["algo"] == "log_surrogate"), **kw, **extra)
    est = SmoothNMF(n_components=c["k"], G=c["G"], shape_2d=c["shape"], algo=c["algo"], tol=0, no_stop_criterion=True, max_iter=4, verbose=0, **kw, **extra)
    with contextlib.redirect_stdout(io.StringIO()):
        est.fit_transform(c["X"], W=c["W0"].copy(), H=c["H0"].copy())
    print(f"{name:16s} ours {np.array(est.losses_)}  oracle {ref['losses']}  store {est._engine.x_store} tile {est._engine.st.tile_px}")
