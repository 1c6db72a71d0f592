#!/usr/bin/env python3
"""How long the estimator's own host copy X_ of a headline-size image takes on this host (base.py's _HostCopy worker): one
thread against a few threads over row blocks (numpy releases the GIL inside copies and ufunc loops)."""
import time
import threading
import numpy as np

X = np.random.default_rng(0).poisson(0.25, size=(2048, 512 * 512)).astype(np.float32)
for rep in range(2):
    t0 = time.perf_counter(); out = X.copy(); t1 = time.perf_counter(); np.multiply(out, 0.5, out=out); t2 = time.perf_counter()
    print(f"one thread: copy {t1 - t0:.3f} s, scale in place {t2 - t1:.3f} s")
    del out
    for nt in (2, 4, 8):
        t0 = time.perf_counter()
        out = np.empty_like(X)
        edges = np.linspace(0, X.shape[0], nt + 1).astype(int)
        th = [threading.Thread(target=lambda a, b: np.multiply(X[a:b], 0.5, out=out[a:b]), args=(edges[i], edges[i + 1])) for i in range(nt)]
        [t.start() for t in th]; [t.join() for t in th]
        print(f"{nt} threads, copy and scale in one pass: {time.perf_counter() - t0:.3f} s")
        del out
