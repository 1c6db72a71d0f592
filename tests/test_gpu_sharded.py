"""Sharded HIP path end to end on ONE GPU: two / three ranks (gloo transport, both on cuda:0) each own a
block of image rows and run the real kernels with halo rows, global statistics and the per-iteration
record exchange; the result must match the unsharded engine.  (RCCL itself needs one GPU per rank; the
driver's multi-GPU bench exercises that transport.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, NX, NY, K, ITERS = 200, 24, 40, 5, 12
KW = dict(lambda_L=1.0, mu=0.1, simplex_H=True, simplex_W=False, tol=0.0)


def _data():
    from espm_amd import synth
    prob = synth.make_problem(N, NX, NY, K, N=40.0, seed=2)
    X = synth.sample_numpy(prob, seed=2)
    X[7] = 0                       # a channel without counts in the whole image, pixels without counts in two shards
    X[:, [3, NY + 1, (NX - 1) * NY + 5]] = 0
    W0, H0 = synth.random_init(N, K, NX * NY, seed=2, scale=0.5)
    return X, W0, H0


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd import sharding
        from espm_amd.engine import MUEngine
        torch.cuda.set_device(0)
        X, W0, H0 = _data()
        row0, rows = sharding.split_rows(NX, world, rank)
        sl = slice(row0 * NY, (row0 + rows) * NY)
        eng = MUEngine(X[:, sl], K, shape_2d=(rows, NY), max_iter=ITERS, group=dist.group.WORLD, device="cuda:0", **KW)
        eng.load_state(W0, H0[:, sl])
        eng.iterate(ITERS, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        out[rank] = (eng.get_W(), eng.get_H(), h["loss"], h["rel_W"], h["rel_H"])
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_engine_matches_single_gpu(world):
    from espm_amd.engine import MUEngine
    X, W0, H0 = _data()
    eng = MUEngine(X, K, shape_2d=(NX, NY), max_iter=ITERS, device="cuda:0", **KW)
    assert eng.x_store == "ell" and eng.st.ell_fill_n == 3    # the sparse store, with its pass for the pixels without counts
    eng.load_state(W0, H0)
    eng.iterate(ITERS, final_loss=True)
    torch.cuda.synchronize()
    ref_W, ref_H, ref = eng.get_W(), eng.get_H(), eng.history()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    for r in range(1, world):
        np.testing.assert_array_equal(res[r][0], res[0][0])       # replicated W bit-identical across ranks
        np.testing.assert_array_equal(res[r][2], res[0][2])       # and so is the assembled loss history
    H = np.concatenate([res[r][1] for r in range(world)], axis=1)
    np.testing.assert_allclose(res[0][0], ref_W, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(H, ref_H, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[0][2], ref["loss"], rtol=1e-6)
    np.testing.assert_allclose(res[0][3][1:], ref["rel_W"][1:], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(res[0][4][1:], ref["rel_H"][1:], rtol=1e-3, atol=1e-6)
