#!/usr/bin/env python3
"""Cost of one iteration on a 1/8 shard of the headline image (64 of 512 rows) on ONE GPU: the C loop
(espm_mu_iterate), the granular host-driven loop the sharded path uses, and a 1-rank RCCL all-gather of the
per-iteration record (host launch cost only: no link is crossed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espm_amd import synth
from espm_amd.engine import MUEngine

ROWS = int(os.environ.get("ROWS", "64"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
prob = synth.make_problem(2048, ROWS, 512, 5, N=500.0, seed=0, row0=0, nx_total=512)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(2048, 5, 512 * 512, seed=0, scale=500.0 / 2048)
eng = MUEngine(X, 5, layout="pm", shape_2d=(ROWS, 512), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=2000, device=dev)
eng.load_state(W0, H0[:, :ROWS * 512])
print("store", eng.x_store, "tile_px", eng.st.tile_px, "nblk_w", eng.st.nblk_w)

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6

eng.iterate(50, final_loss=False)
print("C loop            : host %.1f us/it, total %.1f us/it" % timed(lambda n: eng.iterate(n, final_loss=False), 300))
def granular(n):
    for _ in range(n):
        eng.eval_current(True)
        eng.finish_iteration()
granular(20)
print("granular host loop: host %.1f us/it, total %.1f us/it" % timed(granular, 300))

import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
send = torch.zeros(51200, dtype=torch.uint8, device=dev); recv = torch.zeros(51200, dtype=torch.uint8, device=dev)
def ag(n):
    for _ in range(n):
        dist.all_gather_into_tensor(recv, send)
ag(20)
print("all_gather (1 rank): host %.1f us/call, total %.1f us/call" % timed(ag, 300))
def both(n):
    for _ in range(n):
        eng.eval_current(True)
        eng.finish_iteration()
        dist.all_gather_into_tensor(recv, send)
both(20)
print("granular + gather : host %.1f us/it, total %.1f us/it" % timed(both, 300))
dist.destroy_process_group()
