#!/usr/bin/env python3
"""The one-shot exchange under load on ONE GPU: WORLD ranks (processes) share the device, each with a ROWS x 512 shard of
the 2048-channel headline image - full-width records (320 + 1 reduction workgroups), kernels that fill the chip.  Reports
the waits that gave up and the difference to the unsharded fit.  (Ranks that share a GPU compete for its CUs; on a node
every rank has its own.)"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N, NY, K, ITERS = 2048, 512, 5, 10
WORLD, ROWS = int(os.environ.get("WORLD", "2")), int(os.environ.get("ROWS", "32"))


def data():
    from espm_amd import synth
    nx = WORLD * ROWS
    prob = synth.make_problem(N, nx, NY, K, N=500.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=1000)
    W0, H0 = synth.random_init(N, K, nx * NY, seed=0, scale=500.0 / N)
    return X, W0, H0, nx


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from espm_amd.engine import MUEngine
    torch.cuda.set_device(0)
    X, W0, H0, nx = data()
    sl = slice(rank * ROWS * NY, (rank + 1) * ROWS * NY)
    eng = MUEngine(X[sl], K, layout="pm", shape_2d=(ROWS, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=ITERS + 2,
                   group=dist.group.WORLD, device="cuda:0")
    eng.load_state(W0, H0[:, sl])
    torch.cuda.synchronize()
    import time
    if os.environ.get("STEPWISE"):
        for it in range(ITERS):
            t0 = time.perf_counter()
            eng.iterate(1, final_loss=False)
            torch.cuda.synchronize()
            print(f"  rank {rank} iteration {it}: {(time.perf_counter() - t0) * 1e3:8.2f} ms, waits that gave up so far {eng.exchange.lost_peers()}", flush=True)
        eng.eval_current(False)
    else:
        eng.iterate(ITERS, final_loss=True)
    torch.cuda.synchronize()
    lost = eng.exchange.lost_peers()
    out[rank] = (eng.get_W(), eng.hist[:ITERS + 1].cpu().numpy(), eng.exchange.transport, lost, eng.st.tile_px)
    eng.exchange.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    from espm_amd.engine import MUEngine
    X, W0, H0, nx = data()
    eng = MUEngine(X, K, layout="pm", shape_2d=(nx, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=ITERS + 2)
    eng.load_state(W0, H0)
    eng.iterate(ITERS, final_loss=True)
    torch.cuda.synchronize()
    refW, refl = eng.get_W(), eng.history()["loss"]
    del eng, X
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(worker, args=(WORLD, port, out), nprocs=WORLD, join=True)
        res = dict(out)
    for r in range(WORLD):
        print(f"rank {r}: transport {res[r][2]}, waits that gave up {res[r][3]}, tile_px {res[r][4]}, max |W - W_unsharded| / max W = {np.abs(res[r][0] - refW).max() / refW.max():.2e}")
