import sys
sys.path.insert(0, '.')
import numpy as np, torch
from espm_amd.engine import MUEngine
from oracle import mu_oracle as oc
rng = np.random.default_rng(0)
n,nx,ny,k=64,12,10,5; p=nx*ny
Ht = rng.random((k,p))**2 + 0.05; Ht /= Ht.sum(0, keepdims=True)
Wt = rng.random((n,k))**3*160/n+1e-3
X = rng.poisson(Wt@Ht).astype(np.float64)
W0 = rng.random(Wt.shape)*Wt.mean()*2+1e-3
H0 = rng.random((k,p))+0.05; H0/=H0.sum(0,keepdims=True)
X_ = oc.remove_zeros_lines(X, 1e-14)
for sH in (False, True):
    eng = MUEngine(X, k, simplex_H=sH, simplex_W=False, tol=0, max_iter=3, x_store="f32")
    eng.load_state(W0, H0)
    H1 = eng.step_h_only()
    Href = oc.multiplicative_step_h(X_, np.eye(n), W0, H0, simplex_H=sH)
    e = np.abs(H1-Href); j = np.unravel_index(np.argmax(e), e.shape)
    print("simplex", sH, "max err", e.max(), "at", j, H1[:, j[1]], Href[:, j[1]], "colsum", H1[:, j[1]].sum())
    print("hstat", eng.hstat[0].cpu().numpy(), eng.hstat[1].cpu().numpy(), Href.sum(1), Href.max(1))
    print("colsum_gw", eng.colsum_gw.cpu().numpy(), W0.sum(0))
    print("hist0", eng.hist[0].cpu().numpy())
