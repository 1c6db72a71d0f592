"""Parity with the oracle AT BASELINE.json's full sizes.

The faithful oracle (oracle/mu_oracle.py) forms dense (n, p) fp64 temporaries and a dense identity G: it cannot run at
2048 x 512^2.  What runs here instead, all in fp64 on the host:

* ``oracle/mu_oracle_sparse.py`` - the same update rules evaluated on the non-zero entries of X only, pinned to the
  faithful oracle (and through it to the reference-generated fixture F6) by tests/test_oracle_golden.py::
  test_sparse_oracle_on_f6 - for whole TRAJECTORIES at C3 (2048 x 512 x 512, k = 5, the headline) and C2 (1980 x 128 x 128,
  k = 3);
* direct fp64 evaluations of the update formulas on channel / pixel subsets (an update of a W row needs that channel's
  row of X and all of H; an update of an H column its own column of X, W, four neighbours and the global row maxima) for
  C3's W step and for C5 (1980 x 1024 x 1024, k = 8, G 1980 x 17, mu = 0.05), whose 435 M non-zero entries are too many for
  a host trajectory inside a test.

Tolerances (fp32 device arithmetic against fp64): loss history 1e-5 relative (BASELINE.json's north star), H 5e-5
absolute, W 2e-4 of its scale.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mu_oracle as oc  # noqa: E402
from oracle import mu_oracle_sparse as osp  # noqa: E402

LOSS_RTOL, H_ATOL, W_RTOL = 1e-5, 5e-5, 2e-4


def spread(n, count=64):
    return np.unique(np.linspace(0, n - 1, count).astype(np.int64))


def sparse_from_device(X_pm):
    """SparseX of a (p, n) device matrix of counts (rows = pixels: torch.nonzero walks it pixel by pixel)."""
    nz = torch.nonzero(X_pm)
    val = X_pm[nz[:, 0], nz[:, 1]]
    return osp.SparseX(nz[:, 1].to(torch.int32).cpu().numpy(), nz[:, 0].to(torch.int32).cpu().numpy(),
                       val.double().cpu().numpy(), X_pm.shape[1], X_pm.shape[0])


def check_trajectory(eng, ref, iters, tag):
    hist = eng.history()
    np.testing.assert_allclose(hist["loss"][1:iters + 1], ref["losses"], rtol=LOSS_RTOL, err_msg=tag + " losses")
    np.testing.assert_allclose(hist["loss"][0], ref["eval_init"], rtol=LOSS_RTOL, err_msg=tag + " initial loss")
    np.testing.assert_allclose(hist["rel_W"][1:iters + 1], ref["rel"][:, 0], rtol=2e-3, err_msg=tag + " rel_W")
    np.testing.assert_allclose(hist["rel_H"][1:iters + 1], ref["rel"][:, 1], rtol=2e-3, err_msg=tag + " rel_H")
    W, H = eng.get_W(), eng.get_H()
    np.testing.assert_allclose(H, ref["H"], atol=H_ATOL, rtol=0, err_msg=tag + " H")
    np.testing.assert_allclose(W, ref["W"], atol=W_RTOL * np.abs(ref["W"]).max(), rtol=W_RTOL, err_msg=tag + " W")
    assert hist["bad"].sum() == 0
    return float(np.max(np.abs(hist["loss"][1:iters + 1] - ref["losses"]) / np.abs(ref["losses"])))


# ---- C3: 2048 channels x 512 x 512 pixels, k = 5, simplex_H + Laplacian (the headline) ----------------------------------
N3, NX3, NY3, K3 = 2048, 512, 512, 5


@pytest.fixture(scope="module")
def c3():
    from espm_amd import synth
    prob = synth.make_problem(N3, NX3, NY3, K3, N=500.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=1000)                      # (p, n) counts, as bench.py draws them
    W0, H0 = synth.random_init(N3, K3, NX3 * NY3, seed=0, scale=500.0 / N3)
    return dict(X=X, W0=W0, H0=H0, kw=dict(layout="pm", shape_2d=(NX3, NY3), lambda_L=1.0, simplex_H=True, simplex_W=False,
                                           tol=0.0, max_iter=12))


def test_c3_trajectory_against_the_sparse_oracle(c3):
    """Five full iterations of the headline problem - the fused kernel the benchmark times, and the two-launch path -
    against five iterations of the fp64 oracle on the same X, W0, H0: losses_, rel_, W_, H_."""
    from espm_amd.engine import MUEngine
    iters = 5
    sx = sparse_from_device(c3["X"])
    assert abs(sx.sum_x - float(c3["X"].sum(dtype=torch.float64))) < 1e-6 * sx.sum_x
    ref = osp.fit(sx, K3, W=c3["W0"], H=c3["H0"], lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(NX3, NY3), tol=0.0,
                  max_iter=iters)
    assert osp.dropped_eps_logy(sx, None, ref["W"], ref["H"]) < 1e-9 * abs(ref["losses"][-1]) * N3 * NX3 * NY3
    del sx
    worst = {}
    for fused in (True, False):
        eng = MUEngine(c3["X"], K3, fused=fused, **c3["kw"])
        assert eng.x_store == "ell" and bool(eng.lib.espm_mu_fused_applies(__import__("ctypes").byref(eng.st))) == fused
        eng.load_state(c3["W0"], c3["H0"])
        eng.iterate(iters, final_loss=True)
        torch.cuda.synchronize()
        worst[fused] = check_trajectory(eng, ref, iters, f"C3 fused={fused}")
        np.testing.assert_allclose(eng.get_H().sum(axis=0), 1.0, atol=5e-6)
        del eng
    print("C3 five iterations, worst relative loss error: fused %.2e, two launches %.2e" % (worst[True], worst[False]))


def test_c3_estimator_with_stop_rules_takes_the_same_steps(c3):
    """The estimator's DEFAULT loop - a loss read back and the stop rules evaluated after every iteration, i.e. the
    granular eval_current / finish_iteration sequence with the history read in between - on the fused kernel at the
    headline size: the same iterates as the batch loop (which the sparse oracle checks above)."""
    import contextlib
    import io
    from espm_amd.engine import MUEngine
    from espm_amd.estimators import SmoothNMF
    iters = 3
    eng = MUEngine(c3["X"], K3, **c3["kw"])
    eng.load_state(c3["W0"], c3["H0"])
    eng.iterate(iters, final_loss=True)
    torch.cuda.synchronize()
    ref_W, ref_H, ref_loss = eng.get_W(), eng.get_H(), eng.history()["loss"][1:iters + 1]
    del eng
    Xh = c3["X"].T.contiguous().cpu().numpy()                              # (n, p) float32 on the host, as a user hands it over
    est = SmoothNMF(n_components=K3, shape_2d=(NX3, NY3), lambda_L=1.0, simplex_H=True, simplex_W=False, max_iter=iters, verbose=0, tol=1e-9)
    with contextlib.redirect_stdout(io.StringIO()):
        est.fit_transform(Xh, W=c3["W0"].astype(np.float32), H=c3["H0"].astype(np.float32))
    assert est.n_iter_ == iters and est._engine.x_store == "ell"
    np.testing.assert_allclose(est.losses_, ref_loss, rtol=1e-6)
    np.testing.assert_allclose(est.H_, ref_H, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(est.W_, ref_W, rtol=1e-5, atol=1e-8)


def test_c3_w_rows_in_fp64_on_64_channels(c3):
    """W step at the headline size (32 channel groups, 256 pixel blocks, the slab reduction with the update folded in):
    for 64 channels spread over the spectrum, A[c, :] = sum_j X[c, j] / (GW[c] . H'[:, j]) H'[:, j] and
    W'[c, :] = max(W[c, :] A[c, :] / rowsum(H'), eps) (updates.py:38-39, :53-60, :70-72, G = identity) in fp64 on the host
    from the H' the device produced (itself checked above and in test_gpu_fullsize.py)."""
    from espm_amd.engine import MUEngine
    eng = MUEngine(c3["X"], K3, **c3["kw"])
    eng.load_state(c3["W0"], c3["H0"])
    eng.iterate(1, final_loss=False)
    torch.cuda.synchronize()
    H1 = eng.get_H().astype(np.float64)
    W1 = eng.get_W()
    cs = spread(N3)
    Xc = c3["X"][:, torch.from_numpy(cs).cuda()].double().cpu().numpy().T       # (64, p)
    W0 = c3["W0"]
    Y = W0[cs] @ H1                                                               # (64, p)
    A = (Xc / Y) @ H1.T                                                           # (64, k)
    np.testing.assert_allclose(eng.a.cpu().numpy()[:, cs].T, A, rtol=2e-5, atol=1e-6)
    ref = np.maximum(W0[cs] * A / H1.sum(axis=1)[None, :], 1e-14)
    np.testing.assert_allclose(W1[cs], ref, rtol=2e-5, atol=1e-9)


# ---- C2: 1980 channels x 128 x 128 pixels, k = 3, simplex_H only ---------------------------------------------------------
@pytest.mark.parametrize("store", ["f32", "auto"])
def test_c2_full_size_trajectory(store):
    """BASELINE config 2 at its full size, on the fp32 store the configuration names and on the store the engine picks."""
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    n, nx, ny, k, iters = 1980, 128, 128, 3, 6
    prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=2000)
    W0, H0 = synth.random_init(n, k, nx * ny, seed=0, scale=500.0 / n)
    ref = osp.fit(sparse_from_device(X), k, W=W0, H=H0, lambda_L=0.0, simplex_H=True, simplex_W=False, shape_2d=(nx, ny), tol=0.0,
                  max_iter=iters)
    eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=0.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=iters + 2,
                   x_store=store)
    assert eng.x_store == ("f32" if store == "f32" else "ell")
    eng.load_state(W0, H0)
    eng.iterate(iters, final_loss=True)
    torch.cuda.synchronize()
    check_trajectory(eng, ref, iters, f"C2 {store}")
    assert np.all(np.diff(eng.history()["loss"]) < 0)


# ---- C5: 1980 channels x 1024 x 1024 pixels, k = 8, fixed dictionary G 1980 x 17, mu = 0.05, lambda = 1, simplex_H ---------
def test_c5_full_size_against_fp64_subsets():
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    n, nx, ny, k, m = 1980, 1024, 1024, 8, 17
    prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0, m=m)
    G = prob["G"]
    X = synth.sample_torch(prob, "cuda", seed=3000)                      # (p, n) 8.3 GB of counts
    W0, H0 = synth.random_init(m, k, nx * ny, seed=0, scale=500.0 / n)
    mu = 0.05
    eng = MUEngine(X, k, layout="pm", G=G, shape_2d=(nx, ny), lambda_L=1.0, mu=mu, simplex_H=True, simplex_W=False, tol=0.0, max_iter=8)
    assert eng.x_store == "ell"
    eng.load_state(W0, H0)
    H1 = eng.step_h_only().astype(np.float64)
    # (1) the first H update on three image rows (+ one halo row each side), updates.py:83-156 in fp64
    r0 = 500
    ext = slice((r0 - 1) * ny, (r0 + 4) * ny)
    Xs = X[ext].double().cpu().numpy().T                                 # (n, 5 ny)
    GW = G @ W0
    Hs = H0[:, ext]
    L = oc.laplacian_matrix(5, ny)
    maxH = H0.max(axis=1, keepdims=True)
    num = GW.T @ (Xs / (GW @ Hs)) + 1.0 * 8 * maxH
    den = GW.sum(axis=0)[:, None] + mu / (Hs + 1.0) + 1.0 * 8 * maxH + 1.0 * (Hs @ L)
    num = Hs * num
    delta, e = oc.dichotomy_simplex_exact(num, den)
    ref = np.fmax(num / (delta + e), 1e-14)[:, ny:-ny]
    np.testing.assert_allclose(H1[:, r0 * ny:(r0 + 3) * ny], ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(H1.sum(axis=0), 1.0, atol=5e-6)
    # (2) one full iteration: A = R H'^T on 64 channels in fp64, then W' from the device's A (the finish: G^T A, the
    # denominators colsum(G) rowsum(H'), clamp; updates.py:58-60, :70-72)
    eng.load_state(W0, H0)
    eng.iterate(1, final_loss=False)
    torch.cuda.synchronize()
    H1 = eng.get_H().astype(np.float64)
    cs = spread(n)
    Xc = X[:, torch.from_numpy(cs).cuda()].double().cpu().numpy().T       # (64, p)
    A = (Xc / (GW[cs] @ H1)) @ H1.T
    A_dev = eng.a.cpu().numpy()[:, :n].T.astype(np.float64)                # (n, k)
    np.testing.assert_allclose(A_dev[cs], A, rtol=2e-5, atol=1e-6)
    ref_W = np.maximum(W0 * (G.T @ A_dev) / (G.sum(axis=0)[:, None] * H1.sum(axis=1)[None, :]), 1e-14)
    np.testing.assert_allclose(eng.get_W(), ref_W, rtol=2e-5, atol=1e-9)
    W1 = eng.get_W().astype(np.float64)
    # (3) properties over further iterations: monotone objective, simplex, mass balance of the KL update
    eng.iterate(4, final_loss=True)
    torch.cuda.synchronize()
    h = eng.history()
    assert np.all(np.diff(h["loss"]) < 0) and h["bad"].sum() == 0
    # (4) the loss VALUE (Weak 1: the north star's tolerance is on the loss): SmoothNMF.loss (smooth_nmf.py:457-475,
    # base.py:196-203) of the initial state and of the device's state after one iteration, evaluated once each in fp64 on the
    # host over all 435 M non-zero entries, against the history's first two entries
    sx = sparse_from_device(X)
    Lfull = oc.laplacian_matrix(nx, ny)
    for slot, (Wt, Ht) in enumerate(((W0, H0), (W1, H1))):
        ref_loss, parts = osp.loss(sx, G, Wt, Ht, Lfull, mu=mu, epsilon_reg=1, lambda_L=1.0)
        np.testing.assert_allclose(h["loss"][slot], ref_loss, rtol=LOSS_RTOL, err_msg=f"C5 loss of state {slot}")
        print("C5 loss of state %d: device %.9g, fp64 host %.9g (rel %.2e)" % (slot, h["loss"][slot], ref_loss, abs(h["loss"][slot] - ref_loss) / abs(ref_loss)))
    del sx
    np.testing.assert_allclose(eng.get_H().sum(axis=0), 1.0, atol=5e-6)
    sum_y = float(eng.hist[5, 3].item())
    assert abs(sum_y - eng.sum_x) / eng.sum_x < 1e-5
