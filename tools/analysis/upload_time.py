#!/usr/bin/env python3
"""Host-to-device copy of a headline-size X (2.15 GB fp32, pageable numpy memory): torch's own copy against a double-buffered
copy through pinned staging buffers filled by numpy (GIL released) while the previous chunk is on the bus."""
import time
import numpy as np
import torch

X = np.random.default_rng(0).poisson(0.25, size=(2048, 512 * 512)).astype(np.float32)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)


def plain():
    return torch.from_numpy(X).to(dev)


CH = 64   # rows per chunk: 64 MB
pin = [torch.empty((CH, X.shape[1]), dtype=torch.float32).pin_memory() for _ in range(2)]
pin_np = [p.numpy() for p in pin]
ev = [torch.cuda.Event() for _ in range(2)]


def staged():
    out = torch.empty(X.shape, dtype=torch.float32, device=dev)
    for i, a in enumerate(range(0, X.shape[0], CH)):
        j = i & 1
        ev[j].synchronize()
        b = min(a + CH, X.shape[0])
        np.copyto(pin_np[j][:b - a], X[a:b])
        out[a:b].copy_(pin[j][:b - a], non_blocking=True)
        ev[j].record()
    torch.cuda.synchronize()
    return out


for name, fn in (("torch .to() from pageable memory", plain), ("64 MB pinned double buffer", staged), ("torch .to() from pageable memory", plain)):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del y
    print(f"{name:36s}: " + " ".join(f"{1e3 * t:6.1f}" for t in ts) + " ms")
