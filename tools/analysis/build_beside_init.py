#!/usr/bin/env python3
"""Can the sparse store's build run BESIDE the NNDSVD initialisation?  Both need the whole uploaded X and not each other.  Headline size;
the initialisation on the main thread and stream, MUEngine (whose set-up is the store's build) on a worker thread with a stream of its
own; one after the other for comparison.  Repeated: the caching allocator is warm from the second repetition on."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from espm_amd import init_device as idv, synth
from espm_amd.engine import MUEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, nx, ny, k = 2048, 512, 512, 5
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
Xpm = synth.sample_torch(prob, dev, seed=1000)       # (p, n)
Xd = Xpm.T
Xh = np.empty((n, nx * ny), dtype=np.float32)
nnz = int(torch.count_nonzero(Xpm))
facts = dict(nonneg=True, sum_x=float(Xpm.sum(dtype=torch.float64)), is_count=True, nnz=nnz)


def init():
    return idv.initialize_nmf_device(Xh, k, init="nndsvd", random_state=0, X_device=Xd, X_mean=0.24)


def build(stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        eng = MUEngine(Xpm, k, layout="pm", shape_2d=(nx, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=200, device=dev,
                       x_facts=dict(facts), fix_zero_lines=False)
        torch.cuda.current_stream().synchronize()
    return eng


for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    init(); torch.cuda.synchronize(); t1 = time.perf_counter()
    e = build(); torch.cuda.synchronize(); t2 = time.perf_counter()
    del e
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    out = {}
    th = threading.Thread(target=lambda: out.setdefault("e", build(side)))
    torch.cuda.synchronize(); t3 = time.perf_counter()
    th.start()
    init()
    th.join(); torch.cuda.synchronize(); t4 = time.perf_counter()
    out.clear()
    print(f"rep {rep}: initialisation {1e3 * (t1 - t0):6.2f} ms, engine {1e3 * (t2 - t1):6.2f} ms, one after the other {1e3 * (t2 - t0):6.2f} ms; side by side {1e3 * (t4 - t3):6.2f} ms", flush=True)
