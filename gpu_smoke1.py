import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from espm_amd.engine import MUEngine
from oracle import mu_oracle as oc

def run_case(n, nx, ny, k, m, lam, mu, simplex_H, simplex_W, seed=0, iters=5, store="auto", poisson=True):
    rng = np.random.default_rng(seed)
    p = nx*ny
    Ht = rng.random((k,p))**2 + 0.05; Ht /= Ht.sum(0, keepdims=True)
    if m:
        G = rng.random((n,m))*(rng.random((n,m))<0.4)+0.01
        Wt = rng.random((m,k))*40/n
        D = G@Wt
    else:
        G=None; Wt = rng.random((n,k))**3*160/n+1e-3; D=Wt
    X = rng.poisson(D@Ht).astype(np.float64) if poisson else D@Ht
    W0 = rng.random(Wt.shape)*Wt.mean()*2+1e-3
    H0 = rng.random((k,p))+0.05; H0/=H0.sum(0,keepdims=True)
    ref = oc.fit(X, k, G=G, W=W0.copy(), H=H0.copy(), lambda_L=lam, mu=mu, simplex_H=simplex_H, simplex_W=simplex_W,
                 shape_2d=(nx,ny), tol=0, no_stop_criterion=True, max_iter=iters)
    eng = MUEngine(X, k, G=G, shape_2d=(nx,ny), lambda_L=lam, mu=mu, simplex_H=simplex_H, simplex_W=simplex_W, tol=0, max_iter=iters, x_store=store)
    Xz = oc.remove_zeros_lines(X, 1e-14)
    eng.load_state(W0, H0)
    eng.iterate(iters, final_loss=True)
    torch.cuda.synchronize()
    W, H = eng.get_W(), eng.get_H()
    hist = eng.history()
    # oracle W,H before rescale: only compare when simplex on
    eW = np.max(np.abs(W-ref['W'])/(np.abs(ref['W'])+1e-12*0+ref['W'].mean()*1e-3))
    eH = np.max(np.abs(H-ref['H']))
    lref = np.concatenate([[ref['eval_init']], ref['losses']])
    eL = np.max(np.abs(hist['loss']-lref)/np.abs(lref))
    print(f"n={n} p={p} k={k} m={m} lam={lam} mu={mu} sH={simplex_H} sW={simplex_W} store={eng.x_store} tile={eng.st.tile_px}: relW={eW:.2e} absH={eH:.2e} relLoss={eL:.2e} relWref={np.max(np.abs(hist['rel_W'][1:]-ref['rel'][:,0])):.2e} relHref={np.max(np.abs(hist['rel_H'][1:]-ref['rel'][:,1])):.2e} bad={hist['bad'].sum()}")

run_case(64, 12, 10, 5, None, 1.0, 0, True, False)
run_case(64, 12, 10, 5, None, 0.0, 0, True, False, store="f32")
run_case(60, 10, 12, 4, 9, 1.0, 0.05, True, False)
run_case(32, 6, 6, 3, None, 0.5, 0, False, True)
run_case(100, 20, 20, 3, None, 0.0, 0, True, False, poisson=False)
run_case(1980, 32, 32, 3, None, 0.0, 0, True, False)
run_case(300, 40, 50, 8, 17, 1.0, np.array([0,.1,.2,.3,.4,.5,.6,.7]), True, False)
run_case(2048, 64, 64, 5, None, 1.0, 0, True, False, iters=10)
