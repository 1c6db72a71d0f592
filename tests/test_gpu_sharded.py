"""Sharded HIP path end to end on ONE GPU: two / three ranks (process group: gloo, all on cuda:0) each own a
block of image rows and run the real kernels with halo rows, global statistics and the per-iteration
record exchange; the result must match the unsharded engine AND the oracle on the whole image.

Both transports of the exchange (espm_amd/sharding.py): the library's one-shot P2P exchange (mailboxes mapped through
hipIpc - which also works between processes that share a GPU - flags, bounded waits; the batch loop
espm_mu_iterate_sharded and the granular loop of the stop-criteria path) and the collective (here gloo; RCCL itself
needs one GPU per rank: the driver's multi-GPU bench exercises it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

NX, NY, K, ITERS = 24, 40, 5, 12
# "h": the headline's constraints; "w": the reference's default ones - the simplex over W, whose multipliers come from sums over
# the channels that the sum over the ranks' records leaves (224 channels: a multiple of 32, the many-workgroup update applies)
# "h20", "w24": 17..32 components - the third build of the library (component stride 32: the records' statistics are 64 doubles, two
# passes of the granule polls), the dense 8-bit store
CASES = {"h": (200, dict(lambda_L=1.0, mu=0.1, simplex_H=True, simplex_W=False, tol=0.0)),
         "w": (224, dict(lambda_L=0.5, mu=0.05, simplex_H=False, simplex_W=True, tol=0.0)),
         "h20": (200, dict(lambda_L=1.0, mu=0.1, simplex_H=True, simplex_W=False, tol=0.0)),
         "w24": (224, dict(lambda_L=0.5, mu=0.05, simplex_H=False, simplex_W=True, tol=0.0))}
KS = {"h": K, "w": K, "h20": 20, "w24": 24, "g40": K, "g40k20": 20, "g24k20": 20}
# "g40": a dictionary G of 40 columns - more than the one-launch column form of the W step takes (32): the exchange of G^T A runs in
# w_gxchg_update_kernel, the rows of G W' in a launch of their own (what configuration 5's shards ran until round 5)
CASES["g40"] = (200, dict(lambda_L=1.0, mu=0.1, simplex_H=True, simplex_W=False, tol=0.0))
CASES["g40k20"] = CASES["g24k20"] = CASES["g40"]   # (20 components: the third build, whose statistics the exchange kernels poll in two passes - with either W step of a dictionary)
MS = {"g40": 40, "g40k20": 40, "g24k20": 24}


def _data(case):
    from espm_amd import synth
    n, K = CASES[case][0], KS[case]
    prob = synth.make_problem(n, NX, NY, K, N=40.0, seed=2, m=MS.get(case))
    X = synth.sample_numpy(prob, seed=2)
    X[7] = 0                       # a channel without counts in the whole image, pixels without counts in two shards
    X[:, [3, NY + 1, (NX - 1) * NY + 5]] = 0
    W0, H0 = synth.random_init(MS.get(case, n), K, NX * NY, seed=2, scale=0.5)
    if case[0] == "w":
        W0 /= W0.sum(axis=0, keepdims=True)
    if case in MS:
        return X, W0, H0, prob["G"]
    return X, W0, H0, None


def _worker(rank, world, port, out, transport, granular, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESPM_XCHG=transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd import sharding
        from espm_amd.engine import MUEngine
        torch.cuda.set_device(0)
        X, W0, H0, G = _data(case)
        KW, K = CASES[case][1], KS[case]
        row0, rows = sharding.split_rows(NX, world, rank)
        sl = slice(row0 * NY, (row0 + rows) * NY)
        eng = MUEngine(X[:, sl], K, G=G, shape_2d=(rows, NY), max_iter=ITERS, group=dist.group.WORLD, device="cuda:0", force_sharded=(world == 1), **KW)
        eng.load_state(W0, H0[:, sl])
        if granular:      # the loop of the stop-criteria path: one exchange per call, host-sequenced
            for _ in range(ITERS):
                eng.eval_current(True)
                eng.finish_iteration()
            eng.eval_current(False)
        else:
            eng.iterate(ITERS, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        out[rank] = (eng.get_W(), eng.get_H(), h["loss"], h["rel_W"], h["rel_H"], eng.exchange.transport, eng.exchange.lost_peers())
        eng.exchange.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,transport,granular,case", [(2, "p2p", False, "h"), (3, "p2p", False, "h"), (3, "p2p", True, "h"),
                                                           (2, "collective", False, "h"), (3, "collective", True, "h"),
                                                           (2, "p2p", False, "w"), (3, "collective", True, "w"),
                                                           (2, "p2p", False, "h20"), (3, "collective", True, "h20"), (2, "p2p", True, "h20"),   # (p2p with 20 components: two ranks - three processes time-slicing ONE device through these longer kernels can run a peer into the exchange's bounded wait)
                                                           (2, "p2p", False, "w24"), (2, "p2p", False, "g40"), (3, "collective", True, "g40"),
                                                           (2, "p2p", False, "g24k20"), (2, "collective", False, "g40k20"),
                                                           # a group of ONE rank runs the whole protocol - records posted, granules polled, the statistics' passes - without a
                                                           # peer to wait for: the exchange kernels of every W step, deterministically
                                                           (1, "p2p", False, "h"), (1, "p2p", True, "h20"), (1, "p2p", False, "w24"), (1, "p2p", False, "g40"),
                                                           (1, "p2p", False, "g40k20"), (1, "p2p", True, "g40k20"), (1, "p2p", False, "g24k20")])
# (not here: (2, "p2p", ..., "g40k20") - with TWO processes on ONE device the 801 polling workgroups of one rank's w_gxchg_update_kernel can
#  hold the device while the peer that has to post waits for room: the bounded waits give up (LostPeerError) in one run out of two, as at
#  larger shards of any kind on a shared device - DESIGN.md section 5.  On a device per rank there is nobody to wait behind.)
def test_sharded_engine_matches_single_gpu(world, transport, granular, case):
    from espm_amd.engine import MUEngine
    from oracle import mu_oracle as oc
    X, W0, H0, G = _data(case)
    KW, K = CASES[case][1], KS[case]
    eng = MUEngine(X, K, G=G, shape_2d=(NX, NY), max_iter=ITERS, device="cuda:0", **KW)
    if K <= 16:
        assert eng.x_store == "ell" and eng.st.ell_fill_n == 3    # the sparse store, with its pass for the pixels without counts
    else:
        assert eng.x_store == "f32" and eng.V.KP == 32   # (dense stores only; the lines without counts carry the reference's 1e-14 fill: fp32)
    assert case[0] != "w" or eng.st.n_pad % 32 == 0
    eng.load_state(W0, H0)
    eng.iterate(ITERS, final_loss=True)
    torch.cuda.synchronize()
    ref_W, ref_H, ref = eng.get_W(), eng.get_H(), eng.history()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, transport, granular, case), nprocs=world, join=True)
        res = dict(out)
    assert all(res[r][5] == transport and res[r][6] == 0 for r in range(world)), [(res[r][5], res[r][6]) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(res[r][0], res[0][0])       # replicated W bit-identical across ranks
        np.testing.assert_array_equal(res[r][2], res[0][2])       # and so is the assembled loss history
    H = np.concatenate([res[r][1] for r in range(world)], axis=1)
    np.testing.assert_allclose(res[0][0], ref_W, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(H, ref_H, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[0][2], ref["loss"], rtol=1e-6)
    np.testing.assert_allclose(res[0][3][1:], ref["rel_W"][1:], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(res[0][4][1:], ref["rel_H"][1:], rtol=1e-3, atol=1e-6)
    # and against the oracle on the whole image (the data hold lines without counts: the faithful loop with its 1e-14 fill)
    ora = oc.fit(X, K, G=G, W=W0.copy(), H=H0.copy(), shape_2d=(NX, NY), no_stop_criterion=True, max_iter=ITERS, **KW)
    np.testing.assert_allclose(res[0][2][1:], ora["losses"], rtol=1e-5)
    np.testing.assert_allclose(H, ora["H"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(res[0][0], ora["W"], rtol=2e-4, atol=2e-4 * np.abs(ora["W"]).max())
