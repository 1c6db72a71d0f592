#!/usr/bin/env python3
"""Cost of one iteration on a 1/8 shard of the headline image (64 of 512 rows) on ONE GPU: what one rank of an 8-GPU run
does per iteration.  The unsharded C loop on the shard (no exchange at all) against the SHARDED code path with a group of
one rank - records packed, exchanged, combined - on both transports: the library's one-shot exchange (the rank posts to
its own mailbox: every launch and flag of the real protocol, no link crossed) and the RCCL all-gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from espm_amd import synth
from espm_amd.engine import MUEngine

ROWS = int(os.environ.get("ROWS", "64"))
C5 = os.environ.get("CONFIG") == "c5"    # CONFIG=c5 ROWS=128: a rank's share of BASELINE configuration 5 at 8 GPUs (1980 ch, 1024-pixel rows, k = 8, G 1980 x 17, mu = 0.05)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N_CH, NY, K, M = (1980, 1024, 8, 17) if C5 else (2048, 512, 5, None)
prob = synth.make_problem(N_CH, ROWS, NY, K, N=500.0, seed=0, row0=0, nx_total=NY, m=M)
X = synth.sample_torch(prob, dev, seed=1000, row0=0)
W0, H0 = synth.random_init(M if C5 else N_CH, K, NY * NY, seed=0, scale=500.0 / N_CH)
SIMPLEX_W = os.environ.get("SIMPLEX_W") == "1"   # the simplex over W as well (with G = identity: the many-workgroup multiplier search)
kw = dict(layout="pm", shape_2d=(ROWS, NY), lambda_L=1.0, simplex_H=True, simplex_W=SIMPLEX_W, tol=0.0, max_iter=2000, device=dev)
if C5:
    kw.update(G=prob["G"], mu=0.05)


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def granular(eng):
    def run(n):
        for _ in range(n):
            eng.eval_current(True)
            eng.finish_iteration()
    return run


eng = MUEngine(X, K, **kw)
eng.load_state(W0, H0[:, :ROWS * NY])
print("store", eng.x_store, "tile_px", eng.st.tile_px, "nblk_w", eng.st.nblk_w)
eng.iterate(50, final_loss=False)
print("unsharded C loop               : host %5.1f us/it, total %5.1f us/it" % timed(lambda n: eng.iterate(n, final_loss=False), 300))
del eng
for transport in ("p2p", "collective"):
    os.environ["ESPM_XCHG"] = transport
    eng = MUEngine(X, K, group=dist.group.WORLD, force_sharded=True, **kw)
    assert eng.exchange.transport == transport
    eng.load_state(W0, H0[:, :ROWS * NY])
    eng.iterate(50, final_loss=False)
    print("sharded path, %-10s batch: host %5.1f us/it, total %5.1f us/it" % ((transport,) + timed(lambda n: eng.iterate(n, final_loss=False), 300)))
    granular(eng)(20)
    print("sharded path, %-10s granular host loop: host %5.1f us/it, total %5.1f us/it" % ((transport,) + timed(granular(eng), 300)))
    print("   lost peers:", eng.exchange.lost_peers())
    eng.exchange.close()
    del eng
dist.destroy_process_group()
