"""The sharded HIP path at BASELINE.json's FULL geometries: configuration 4 - the headline image,
2048 channels x 512 x 512 pixels, k = 5, simplex_H + Laplacian, cut into 256-row shards (the fused kernel on 512-pixel blocks +
espm_mu_shard_combine_finish) - and the shard configuration 5 names - 128 image rows of 1024 pixels per rank, 1980
channels, k = 8, a fixed dictionary G 1980 x 17, mu = 0.05 (the one-workgroup W finish behind the combine).

Two ranks (and, round 4, four - 128-row shards - as processes and eight - 64-row shards - as threads, tests/thread_ranks.py)
share the one GPU of the box on the collective transport (gloo carries the records; RCCL needs a GPU per rank, and
the one-shot exchange dead-locks between processes that share a device once their workgroups fill it: DESIGN.md section 5).
Checked: W bit-identical across the ranks, W / H against the unsharded engine, the loss history against the fp64 sparse
oracle on the whole image (C4), fp64 evaluations of the update formulas on pixel rows that straddle the shard boundary and on
64 channels (C5 shard), and the properties of the iteration."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

C4 = dict(n=2048, nx=512, ny=512, k=5, m=None, counts=500.0, seed_x=1000, iters=3,
          kw=dict(lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0))
# two of the eight 128-row shards of configuration 5 (a 256 x 1024 image with the same phases, dose and constraints)
C5S = dict(n=1980, nx=256, ny=1024, k=8, m=17, counts=500.0, seed_x=3000, iters=3,
           kw=dict(lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False, tol=0.0))


def _problem(c, row0=0, rows=None):
    from espm_amd import synth
    rows = c["nx"] if rows is None else rows
    prob = synth.make_problem(c["n"], rows, c["ny"], c["k"], N=c["counts"], seed=0, m=c["m"], row0=row0, nx_total=c["nx"])
    X = synth.sample_torch(prob, "cuda", seed=c["seed_x"], row0=row0)      # (p_local, n) counts; the same pixels whoever draws them
    M = c["m"] if c["m"] else c["n"]
    W0, H0 = synth.random_init(M, c["k"], c["nx"] * c["ny"], seed=0, scale=c["counts"] / c["n"])
    return prob, X, W0, H0


def _worker(rank, world, port, out, c):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESPM_XCHG="collective")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd import sharding
        from espm_amd.engine import MUEngine
        torch.cuda.set_device(0)
        row0, rows = sharding.split_rows(c["nx"], world, rank)
        prob, X, W0, H0 = _problem(c, row0, rows)
        sl = slice(row0 * c["ny"], (row0 + rows) * c["ny"])
        eng = MUEngine(X, c["k"], layout="pm", G=prob["G"], shape_2d=(rows, c["ny"]), max_iter=c["iters"] + 2, group=dist.group.WORLD,
                       device="cuda:0", **c["kw"])
        del X
        import ctypes
        fused = bool(eng.lib.espm_mu_fused_applies(ctypes.byref(eng.st)))
        eng.load_state(W0, H0[:, sl])
        eng.iterate(1, final_loss=False)
        torch.cuda.synchronize()
        first = (eng.get_W(), eng.get_H(), eng.a.cpu().numpy())
        eng.iterate(c["iters"] - 1, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        out[rank] = dict(W=eng.get_W(), H=eng.get_H(), loss=h["loss"], rel_W=h["rel_W"], rel_H=h["rel_H"], bad=float(h["bad"].sum()),
                         transport=eng.exchange.transport, store=eng.x_store, fused=fused, tile_px=int(eng.st.tile_px), first=first)
        eng.exchange.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_sharded(c, world=2):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, c), nprocs=world, join=True)
        return {r: dict(out[r]) for r in range(world)}


@pytest.fixture(scope="module")
def c4_reference():
    """What every shard geometry of configuration 4 is compared with, computed once: the unsharded engine on the whole image and
    the fp64 oracle on its non-zero entries (pinned to the reference's F6 by tests/test_oracle_golden.py)."""
    import ctypes
    from espm_amd.engine import MUEngine
    from oracle import mu_oracle_sparse as osp
    from test_gpu_fullsize_parity import sparse_from_device
    c = C4
    prob, X, W0, H0 = _problem(c)
    eng = MUEngine(X, c["k"], layout="pm", shape_2d=(c["nx"], c["ny"]), max_iter=c["iters"] + 2, device="cuda:0", **c["kw"])
    assert bool(eng.lib.espm_mu_fused_applies(ctypes.byref(eng.st)))
    eng.load_state(W0, H0)
    eng.iterate(c["iters"], final_loss=True)
    torch.cuda.synchronize()
    one = dict(hist=eng.history(), W=eng.get_W(), H=eng.get_H())
    del eng
    ora = osp.fit(sparse_from_device(X), c["k"], W=W0, H=H0, shape_2d=(c["nx"], c["ny"]), max_iter=c["iters"], **c["kw"])
    del X
    torch.cuda.empty_cache()
    return dict(one=one, ora=ora)


def _check_config4(res, ref, tile_px):
    """`res`: rank -> result of a sharded run of configuration 4; every rank on the sparse store and the fused launch at H tiles of
    `tile_px` pixels (the block geometry of that shard size), collective transport."""
    world = len(res)
    assert all(res[r]["transport"] == "collective" and res[r]["store"] == "ell" and res[r]["fused"] and res[r]["tile_px"] == tile_px
               and res[r]["bad"] == 0 for r in res), {r: (res[r]["transport"], res[r]["store"], res[r]["fused"], res[r]["tile_px"]) for r in res}
    for r in range(1, world):
        np.testing.assert_array_equal(res[r]["W"], res[0]["W"])           # W is replicated: the same bits on every rank
        np.testing.assert_array_equal(res[r]["loss"], res[0]["loss"])
    H = np.concatenate([res[r]["H"] for r in sorted(res)], axis=1)
    one, ora = ref["one"], ref["ora"]
    np.testing.assert_allclose(res[0]["W"], one["W"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(H, one["H"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[0]["loss"], one["hist"]["loss"], rtol=1e-6)
    np.testing.assert_allclose(res[0]["rel_W"][1:], one["hist"]["rel_W"][1:], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(res[0]["rel_H"][1:], one["hist"]["rel_H"][1:], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(res[0]["loss"][1:], ora["losses"], rtol=1e-5)
    np.testing.assert_allclose(res[0]["loss"][0], ora["eval_init"], rtol=1e-5)
    np.testing.assert_allclose(H, ora["H"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(res[0]["W"], ora["W"], rtol=2e-4, atol=2e-4 * np.abs(ora["W"]).max())
    np.testing.assert_allclose(H.sum(axis=0), 1.0, atol=5e-6)


def test_config4_256_row_shards_of_the_headline_image(c4_reference):
    # (a 256-row shard fills the chip with 512-pixel blocks - two H tiles of 256: the fused launch at its run-time geometry)
    _check_config4(_run_sharded(C4, world=2), c4_reference, tile_px=256)


def test_config4_128_row_shards_four_ranks(c4_reference):
    """BASELINE configuration 4 at FOUR GPUs: four processes with 128-row shards of the 2048 x 512^2 image share
    the one GPU - the fused launch on 256-pixel blocks (H tiles of 128), four records summed in rank order."""
    _check_config4(_run_sharded(C4, world=4), c4_reference, tile_px=128)


def _thread_rank(c):
    def body(group, rank):
        import ctypes
        from espm_amd import sharding
        from espm_amd.engine import MUEngine
        world = group.shared.world
        row0, rows = sharding.split_rows(c["nx"], world, rank)
        prob, X, W0, H0 = _problem(c, row0, rows)
        sl = slice(row0 * c["ny"], (row0 + rows) * c["ny"])
        eng = MUEngine(X, c["k"], layout="pm", G=prob["G"], shape_2d=(rows, c["ny"]), max_iter=c["iters"] + 2, group=group,
                       device="cuda:0", **c["kw"])
        del X
        fused = bool(eng.lib.espm_mu_fused_applies(ctypes.byref(eng.st)))
        eng.load_state(W0, H0[:, sl])
        eng.iterate(c["iters"], final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        return dict(W=eng.get_W(), H=eng.get_H(), loss=h["loss"], rel_W=h["rel_W"], rel_H=h["rel_H"], bad=float(h["bad"].sum()),
                    transport=eng.exchange.transport, store=eng.x_store, fused=fused, tile_px=int(eng.st.tile_px))
    return body


def test_config4_64_row_shards_eight_ranks(c4_reference, monkeypatch):
    """BASELINE configuration 4 at EIGHT GPUs: 64-row shards - the run-time-sized fused instance on 128-pixel blocks (H tiles of 64)
    at the full width of 2048 channels, eight records per sum.  The box admits six processes on its card, so the eight ranks
    are THREADS of this process (tests/thread_ranks.py): each with its own engine, the collective calls of the engine and of
    espm_amd.sharding served by a rendezvous in rank order; the kernels and the C-ABI sequence are the ones eight processes run."""
    from thread_ranks import run_ranks
    monkeypatch.setenv("ESPM_XCHG", "collective")
    res = dict(enumerate(run_ranks(8, _thread_rank(C4))))
    _check_config4(res, c4_reference, tile_px=64)


def test_thread_ranks_agree_with_process_ranks(c4_reference, monkeypatch):
    """The thread harness against real processes on the same geometry (two 256-row shards): identical bits."""
    from thread_ranks import run_ranks
    monkeypatch.setenv("ESPM_XCHG", "collective")
    thr = run_ranks(2, _thread_rank(C4))
    prc = _run_sharded(C4, world=2)
    for r in range(2):
        np.testing.assert_array_equal(thr[r]["W"], prc[r]["W"])
        np.testing.assert_array_equal(thr[r]["H"], prc[r]["H"])
        np.testing.assert_array_equal(thr[r]["loss"], prc[r]["loss"])


def test_config5_128_row_shards():
    from espm_amd.engine import MUEngine
    from oracle import mu_oracle as oc
    from test_gpu_fullsize_parity import spread
    c = C5S
    n, nx, ny, k, mu = c["n"], c["nx"], c["ny"], c["k"], c["kw"]["mu"]
    res = _run_sharded(c)
    assert all(res[r]["transport"] == "collective" and res[r]["store"] == "ell" and res[r]["bad"] == 0 for r in res)
    np.testing.assert_array_equal(res[1]["W"], res[0]["W"])
    np.testing.assert_array_equal(res[1]["loss"], res[0]["loss"])
    np.testing.assert_array_equal(res[1]["first"][2], res[0]["first"][2])   # the summed A too
    H = np.concatenate([res[r]["H"] for r in sorted(res)], axis=1)
    prob, X, W0, H0 = _problem(c)
    G = prob["G"]
    # (1) the first H update in fp64 (updates.py:83-156) on the four image rows around the shard boundary (rows 126..129:
    #     rows 127 | 128 take their lower | upper neighbours from the OTHER rank's boundary row) - the global row maxima of H0
    H1 = np.concatenate([res[r]["first"][1] for r in sorted(res)], axis=1).astype(np.float64)
    r0 = nx // 2 - 2
    ext = slice((r0 - 1) * ny, (r0 + 5) * ny)
    Xs = X[ext].double().cpu().numpy().T
    GW = G @ W0
    Hs = H0[:, ext]
    L = oc.laplacian_matrix(6, ny)
    maxH = H0.max(axis=1, keepdims=True)
    num = GW.T @ (Xs / (GW @ Hs)) + 1.0 * 8 * maxH
    den = GW.sum(axis=0)[:, None] + mu / (Hs + 1.0) + 1.0 * 8 * maxH + 1.0 * (Hs @ L)
    num = Hs * num
    delta, e = oc.dichotomy_simplex_exact(num, den)
    refH = np.fmax(num / (delta + e), 1e-14)[:, ny:-ny]
    np.testing.assert_allclose(H1[:, r0 * ny:(r0 + 4) * ny], refH, rtol=2e-5, atol=2e-6)
    # (2) A = R H'^T on 64 channels in fp64 over BOTH shards against the ranks' summed A, then W' from it (G^T A, colsum(G) rowsum(H'))
    cs = spread(n)
    Xc = X[:, torch.from_numpy(cs).cuda()].double().cpu().numpy().T
    A = (Xc / (GW[cs] @ H1)) @ H1.T
    A_dev = res[0]["first"][2][:, :n].T.astype(np.float64)
    np.testing.assert_allclose(A_dev[cs], A, rtol=2e-5, atol=1e-6)
    ref_W = np.maximum(W0 * (G.T @ A_dev) / (G.sum(axis=0)[:, None] * H1.sum(axis=1)[None, :]), 1e-14)
    np.testing.assert_allclose(res[0]["first"][0], ref_W, rtol=2e-5, atol=1e-9)
    # (3) against the unsharded engine on the 256-row image, and the properties of the iteration
    eng = MUEngine(X, k, layout="pm", G=G, shape_2d=(nx, ny), max_iter=c["iters"] + 2, device="cuda:0", **c["kw"])
    eng.load_state(W0, H0)
    eng.iterate(c["iters"], final_loss=True)
    torch.cuda.synchronize()
    ref = eng.history()
    np.testing.assert_allclose(res[0]["W"], eng.get_W(), rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(H, eng.get_H(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[0]["loss"], ref["loss"], rtol=1e-6)
    assert np.all(np.diff(res[0]["loss"]) < 0)
    np.testing.assert_allclose(H.sum(axis=0), 1.0, atol=5e-6)
    # (4) the whole trajectory - loss values included (Weak 1) - against the fp64 oracle on the image's non-zero entries
    from oracle import mu_oracle_sparse as osp
    from test_gpu_fullsize_parity import sparse_from_device
    ora = osp.fit(sparse_from_device(X), k, G=G, W=W0, H=H0, shape_2d=(nx, ny), max_iter=c["iters"], **c["kw"])
    np.testing.assert_allclose(res[0]["loss"][1:], ora["losses"], rtol=1e-5)
    np.testing.assert_allclose(res[0]["loss"][0], ora["eval_init"], rtol=1e-5)
    np.testing.assert_allclose(H, ora["H"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(res[0]["W"], ora["W"], rtol=2e-4, atol=2e-4 * np.abs(ora["W"]).max())
