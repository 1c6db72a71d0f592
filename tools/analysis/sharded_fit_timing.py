#!/usr/bin/env python3
"""A whole fit sharded over WORLD ranks that share the one GPU of the build box (process group gloo, collective transport), with
ESPM_FIT_TIMING=1: every rank prints the host-side time stamps of its fit's sections and its peak device memory - the upload, the
scans and the initialisation are per BLOCK of image rows since round 4 (espm_amd/estimators/base.py, init_device.randomized_svd_sharded).
WORLD=1 runs the same fit unsharded for comparison.  ROWS x 512 pixels x 2048 channels, k = 5, 200 iterations.

    WORLD=2 python tools/analysis/sharded_fit_timing.py"""
import contextlib
import io
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

WORLD, ROWS = int(os.environ.get("WORLD", "2")), int(os.environ.get("ROWS", "512"))


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESPM_XCHG="collective", ESPM_FIT_TIMING="1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from espm_amd import synth
    from espm_amd.estimators import SmoothNMF
    prob = synth.make_problem(2048, ROWS, 512, 5, N=500.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=1000).t().contiguous().cpu().numpy()      # (n, p) fp32 on the host, as a caller hands it over
    torch.cuda.empty_cache()
    for rep in range(3):
        est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(ROWS, 512), max_iter=200, tol=0, no_stop_criterion=True,
                        verbose=0, random_state=0)
        if world > 1:
            est.shard(dist.group.WORLD)
            dist.barrier()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            est.fit_transform(X)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        peak = (torch.cuda.max_memory_allocated() - base) / 2 ** 20
        marks = [ln for ln in buf.getvalue().splitlines() if ln.startswith("[fit timing")]
        print(f"rank {rank} of {world}, fit {rep}: {dt:.3f} s, peak device memory {peak:.0f} MiB, final loss {est.losses_[-1]:.8f}\n    " + "\n    ".join(marks), flush=True)
        del est
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if WORLD == 1:
        worker(0, 1, port)
    else:
        mp.spawn(worker, args=(WORLD, port), nprocs=WORLD, join=True)
