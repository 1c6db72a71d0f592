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
                                                   (3, "collective", "linesearch_truth"), (2, "p2p", "physics"), (2, "collective", "misc"), (2, "p2p", "pg")])
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
