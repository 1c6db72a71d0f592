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


NB = 4
pin4 = [torch.empty((CH, X.shape[1]), dtype=torch.float32).pin_memory() for _ in range(NB)]
pin4_np = [p.numpy() for p in pin4]
ev4 = [torch.cuda.Event() for _ in range(NB)]


def staged_threads(nthreads=2):
    # `nthreads` host threads fill the pinned buffers (numpy releases the GIL), each thread owns the buffers j = t mod nthreads
    import threading
    out = torch.empty(X.shape, dtype=torch.float32, device=dev)
    chunks = list(range(0, X.shape[0], CH))
    streams = [torch.cuda.Stream() for _ in range(nthreads)]

    def work(t):
        with torch.cuda.stream(streams[t]):
            for i in range(t, len(chunks), nthreads):
                j = (i // nthreads % (NB // nthreads)) * nthreads + t
                ev4[j].synchronize()
                a = chunks[i]
                b = min(a + CH, X.shape[0])
                np.copyto(pin4_np[j][:b - a], X[a:b])
                out[a:b].copy_(pin4[j][:b - a], non_blocking=True)
                ev4[j].record(streams[t])
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    return out


for name, fn in (("torch .to() from pageable memory", plain), ("64 MB pinned double buffer", staged), ("pinned buffers filled by 2 threads", staged_threads),
                 ("torch .to() from pageable memory", plain)):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if ts[-1:] and len(ts) == 1:
            assert bool((y.cpu() == torch.from_numpy(X)).all()), name
        del y
    print(f"{name:36s}: " + " ".join(f"{1e3 * t:6.1f}" for t in ts) + " ms")
