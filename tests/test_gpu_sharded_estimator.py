"""One fit sharded over several ranks THROUGH THE ESTIMATOR (``SmoothNMF(...).shard(group)``, espm_amd/estimators/base.py):
the golden tests of tests/test_gpu_estimator.py - the reference's own trajectories, stop rules, linesearch decisions, truth
tracking and physics-model refreshes - are run again inside 2 / 3 ranks (process group gloo, all ranks on cuda:0, real
kernels, both transports of the record exchange), every rank checking the WHOLE result against the reference's fixture;
the ranks' results are then compared with each other bit for bit."""
import contextlib
import datetime
import io
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

G = "<the golden loader>"
# (test function of tests/test_gpu_estimator.py, its arguments after SmoothNMF)
CASES = {
    "trajectories": [("test_trajectories_golden", (G, name, "auto")) for name in ("c1", "c2", "c3", "c5", "cw")],
    "linesearch_truth": [("test_linesearch_and_truth_tracking_golden", (G, "auto"))],
    "physics": [("test_physics_model_trajectories_golden", (G, "auto"))],
    "misc": [("test_iteration_method_matches_oracle", (G, "auto")),
             ("test_empty_channels_keep_the_sparse_store", (dict(simplex_H=True, simplex_W=False, mu=0.3, lambda_L=2.0), 5, "auto")),
             ("test_empty_channels_keep_the_sparse_store", (dict(simplex_H=False, simplex_W=True), None, "auto")),
             ("test_bregman_variant_golden", (G, "auto")), ("test_frobenius_fit_golden", (G,))],
    "pg": [("test_projected_gradient_golden", (G,)), ("test_projected_gradient_linesearch_golden", (G,))],
    # 17..32 components: whole fits (NNDSVD, default stop rule) on the third build of the library, sharded
    "wide32": [("test_whole_fit_with_more_than_sixteen_components", (20, True)), ("test_whole_fit_with_more_than_sixteen_components", (32, False))],
}


def _golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]
    return load


def _worker(rank, world, port, out, transport, group):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESPM_XCHG=transport)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        torch.cuda.set_device(0)
        from espm_amd.estimators import SmoothNMF
        for path in (HERE, os.path.dirname(HERE)):   # (the test modules import their helpers and the oracle by top-level name)
            if path not in sys.path:
                sys.path.insert(0, path)
        import test_gpu_estimator as T
        made = []

        class Sharded(SmoothNMF):   # (a class: some of the tests derive from what they are handed)
            def fit_transform(self, X, y=None, W=None, H=None):
                self.shard(dist.group.WORLD)
                if not any(e is self for e in made):
                    made.append(self)
                return super().fit_transform(X, y=y, W=W, H=H)
        golden = _golden()
        errors, summary = [], []
        for fn, args in CASES[group]:
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    getattr(T, fn)(Sharded, *[golden if a is G else a for a in args])
            except BaseException:  # noqa: BLE001 - reported to the parent, which fails the test
                errors.append(f"{fn}{args}: {traceback.format_exc()}")
                break   # (every rank checks the same whole-image results, so they stop at the same case)
        for est in made:
            eng = getattr(est, "_engine", None)
            if eng is not None and hasattr(est, "losses_"):
                summary.append((np.asarray(est.W_).copy(), np.asarray(est.H_).copy(), np.asarray(est.losses_).copy(), eng.world,
                                eng.exchange.transport, eng.exchange.lost_peers()))
        out[rank] = (errors, summary)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,transport,group", [(2, "p2p", "trajectories"), (3, "collective", "trajectories"), (2, "p2p", "linesearch_truth"),
                                                   (3, "collective", "linesearch_truth"), (2, "p2p", "physics"), (2, "collective", "misc"), (2, "p2p", "pg"),
                                                   (2, "collective", "wide32"), (2, "p2p", "wide32")])
def test_sharded_estimator_reproduces_the_golden_fits(world, transport, group):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, transport, group), nprocs=world, join=True)
        res = dict(out)
    for r in range(world):
        assert not res[r][0], f"rank {r}: " + "\n".join(res[r][0])
    fits = res[0][1]
    assert fits and all(len(res[r][1]) == len(fits) for r in range(world))
    for i, (W, H, losses, w, tr, lost) in enumerate(fits):
        assert w == world and tr == transport and lost == 0
        for r in range(1, world):
            np.testing.assert_array_equal(res[r][1][i][0], W)        # replicated W: bit-identical on all ranks
            np.testing.assert_array_equal(res[r][1][i][1], H)        # the assembled H too
            np.testing.assert_array_equal(res[r][1][i][2], losses)   # and the loss history (same stop decisions)


# ---- a LARGE X (the device-prep path of fit_transform): shard-local upload, scans and initialisation ----
def _large_problem(n, nx, ny, k, holes):
    from espm_amd import synth
    prob = synth.make_problem(n, nx, ny, k, N=200.0, seed=5)
    X = np.ascontiguousarray(synth.sample_numpy(prob, seed=11).astype(np.float32))            # (n, p) counts, C order
    if holes:   # channels and pixels without a single count: filled with log_shift (base.py:519-528), jointly over the ranks
        X[3, :] = 0
        X[n - 2, :] = 0
        X[:, 5] = 0
        X[:, nx * ny - 7] = 0          # (one empty pixel in each rank's block)
    return X


def _fit_large(X, k, nx, ny, group, hspy):
    from espm_amd.estimators import SmoothNMF
    est = SmoothNMF(n_components=k, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(nx, ny), max_iter=25, tol=0, no_stop_criterion=True,
                    verbose=0, random_state=0, hspy_comp=hspy)
    if group is not None:
        est.shard(group)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with contextlib.redirect_stdout(io.StringIO()):
        est.fit_transform(np.ascontiguousarray(X.T) if hspy else X)   # (hyperspy hands over a C-ordered (pixels, channels) matrix)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    return dict(W=np.asarray(est.W_), H=np.asarray(est.H_), losses=np.asarray(est.losses_), peak=int(peak), store=est._engine.x_store,
                layout=est._ingest_layout, X_sum=float(np.asarray(est.X_, dtype=np.float64).sum()), X_min=float(np.asarray(est.X_).min()), const_KL=float(est.const_KL_))


def _large_worker(rank, world, port, out, cfg):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESPM_XCHG="collective")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        torch.cuda.set_device(0)
        n, nx, ny, k, holes, hspy = cfg
        X = _large_problem(n, nx, ny, k, holes)
        res = _fit_large(X, k, nx, ny, dist.group.WORLD, hspy)
        # the sharded randomized SVD against the one-GPU routine on the whole image (same random stream)
        from espm_amd import init_device
        from espm_amd.estimators.base import _Shard
        sh = _Shard(dist.group.WORLD, (nx, ny), nx * ny)
        Xd = torch.from_numpy(X).cuda()
        U1, s1, V1 = init_device.randomized_svd_device(Xd, k, 0)
        U2, s2, V2 = init_device.randomized_svd_sharded(Xd[:, sh.sl].contiguous(), k, 0, sh)
        res["svd"] = (float(np.abs(s2 - s1).max() / s1.max()), float(np.abs(U2 - U1).max()), float(np.abs(V2 - V1).max() / np.abs(V1).max()))
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [(512, 256, 256, 4, False, False), (384, 128, 128, 3, True, False), (384, 128, 128, 3, False, True)],
                         ids=["counts-2ranks", "holes", "pixel-major"])
def test_large_x_is_ingested_and_initialised_shard_locally(cfg):
    """A fit whose X takes the device-prep path (>= 4 M entries), sharded over two ranks: every rank uploads, scans and initialises on ITS
    block of image rows only - the same fit as on one GPU to the rounding of another order of summation (NNDSVD from a randomized SVD whose
    sums over the pixels are all-reduced), W bit-identical across the ranks, and a peak of device memory per rank that follows its share."""
    n, nx, ny, k, holes, hspy = cfg
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_large_worker, args=(world, _free_port(), out, cfg), nprocs=world, join=True)
        res = {r: dict(out[r]) for r in range(world)}
    X = _large_problem(n, nx, ny, k, holes)
    one = _fit_large(X, k, nx, ny, None, hspy)
    for r in range(1, world):
        np.testing.assert_array_equal(res[r]["W"], res[0]["W"])
        np.testing.assert_array_equal(res[r]["H"], res[0]["H"])
        np.testing.assert_array_equal(res[r]["losses"], res[0]["losses"])
    assert res[0]["store"] == one["store"] and res[0]["layout"] == one["layout"] == ("pm" if hspy else "cm")
    ds, du, dv = res[0]["svd"]
    assert ds < 1e-5 and du < 2e-4 and dv < 2e-4, res[0]["svd"]
    np.testing.assert_allclose(res[0]["const_KL"], one["const_KL"], rtol=1e-12)
    assert res[0]["X_sum"] == one["X_sum"] and res[0]["X_min"] == one["X_min"]      # the estimator's own copy of the data, empty lines filled
    np.testing.assert_allclose(res[0]["losses"], one["losses"], rtol=2e-5)
    # (two initialisations that agree to fp32 rounding, then the iterations: a handful of the 65 k - 262 k pixels sit where a component is
    #  handed from one neighbour to the other and amplify that rounding)
    dH = np.abs(res[0]["H"] - one["H"])
    assert dH.max() < 1e-2 and (dH > 2e-3).mean() < 2e-5, (dH.max(), int((dH > 2e-3).sum()))
    np.testing.assert_allclose(res[0]["W"], one["W"], rtol=5e-3, atol=5e-3 * np.abs(one["W"]).max())
    if n * nx * ny >= 32 << 20:   # where the image dominates the peak (the 25 MB cases sit under the BLAS workspace torch allocates per process): a rank's follows its share (X as uploaded fp32 + the store's build), not the image
        assert max(res[r]["peak"] for r in res) <= 0.65 * one["peak"], (res[0]["peak"], res[1]["peak"], one["peak"])
