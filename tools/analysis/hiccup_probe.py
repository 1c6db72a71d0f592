#!/usr/bin/env python3
"""Is there something PERIODIC on the box that stops the device or the host thread for tens of milliseconds?
Three probes of 6 s each, every gap above 5 ms printed with its time stamp:
  host   : a thread that does nothing but read the clock (descheduling, CPU steal)
  device : tiny kernel + synchronize in a tight loop (a stalled queue or a late completion signal)
  copy   : a 64 MB host-to-device copy from pinned memory in a loop (the path an upload takes)"""
import sys, time
import torch

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
torch.cuda.synchronize()


def probe(name, body, thresh):
    t_start = time.perf_counter()
    last = t_start
    n = 0
    gaps = []
    while last - t_start < SECONDS:
        body()
        now = time.perf_counter()
        if now - last > thresh:
            gaps.append((last - t_start, now - last))
        last = now
        n += 1
    print(f"{name}: {n} rounds in {SECONDS:.0f} s ({1e6 * SECONDS / n:.1f} us each); {len(gaps)} gaps above {1e3 * thresh:.0f} ms")
    for at, d in gaps[:60]:
        print(f"    at {at:7.3f} s: {1e3 * d:7.1f} ms")


probe("host  ", lambda: None, 5e-3)


def dev_body():
    x.add_(1.0)
    torch.cuda.synchronize()


probe("device", dev_body, 5e-3)
pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def copy_body():
    dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()


probe("copy  ", copy_body, 5e-3)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def stream_body():   # 1 GB read + write on the device: 2 GB of HBM traffic per round
    big.add_(1)
    torch.cuda.synchronize()


probe("hbm   ", stream_body, 5e-3)
