#!/usr/bin/env python3
"""cProfile of initialize_algorithms at the headline size with X already on the device (what a fit's initialisation costs on the host)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from espm_amd._cpu_budget import limited_thread_pools
from espm_amd.estimators.updates import initialize_algorithms
n, p, k = 2048, 512 * 512, 5
dev = torch.device("cuda", 0)
Xd = torch.poisson(torch.full((n, p), 0.25, device=dev))
Xh = np.broadcast_to(np.float32(0), (n, p))      # (shape and dtype only: X_device carries the data)
with limited_thread_pools():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pr = cProfile.Profile(); pr.enable()
        G, W, H = initialize_algorithms(Xh, None, None, None, k, "nndsvdar", 0, True, False, X_device=Xd)
        pr.disable(); torch.cuda.synchronize()
        print(f"rep {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
