#!/usr/bin/env python3
"""What first-touching a 2 GB host array costs on this box, and what it does to another thread's small allocations (the
estimator's host copy X_ is made by worker threads while the main thread drives the fit): transparent huge pages or not,
one thread or four, and the latency of 1 MB numpy allocations in the main thread meanwhile."""
import threading, time
import numpy as np
for f in ("enabled", "defrag"):
    try:
        print("THP", f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as e:
        print("THP", f, e)


def touch(n_threads, shape=(2048, 262144)):
    a = np.empty(shape, np.float32)
    edges = np.linspace(0, shape[0], n_threads + 1).astype(int)
    ts = [threading.Thread(target=lambda lo, hi: a[lo:hi].fill(1.0), args=(int(edges[i]), int(edges[i + 1]))) for i in range(n_threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    lat = []
    while any(t.is_alive() for t in ts):      # the main thread meanwhile: small allocations, as torch / numpy temporaries make them
        s = time.perf_counter()
        b = np.empty(1 << 18, np.float32)
        b[::1024] = 0
        del b
        lat.append(time.perf_counter() - s)
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    lat = np.array(lat) * 1e3
    print(f"first touch of {a.nbytes / 1e9:.1f} GB by {n_threads} thread(s): {dt * 1e3:.0f} ms; main thread's 1 MB allocations meanwhile: "
          f"{len(lat)} done, median {np.median(lat):.3f} ms, p99 {np.percentile(lat, 99):.2f} ms, max {lat.max():.2f} ms")
    t0 = time.perf_counter()
    a.fill(2.0)
    print(f"   second pass (pages present), one thread: {(time.perf_counter() - t0) * 1e3:.0f} ms")


for n in (1, 4, 4):
    touch(n)
