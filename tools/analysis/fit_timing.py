#!/usr/bin/env python3
"""Five fits of the headline problem (200 iterations, host fp32 array in, host arrays out) with the estimator's own host-side
time stamps (ESPM_FIT_TIMING=1: no device synchronisation added): which section of fit_transform the time sits in."""
import os, sys, time
os.environ["ESPM_FIT_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, io
import numpy as np
import torch
from espm_amd import synth
from espm_amd.estimators import SmoothNMF

import gc
if os.environ.get("GC") == "off":       # does the ~75 ms that wander between the sections belong to the interpreter's collector?
    gc.disable()
elif os.environ.get("GC") == "log":
    _t = {}
    def _cb(phase, info):
        if phase == "start":
            _t["t"] = time.perf_counter()
        else:
            dt = time.perf_counter() - _t["t"]
            if dt > 2e-3:
                print(f"   [gc] generation {info['generation']}: {1e3 * dt:.1f} ms, collected {info['collected']}", file=sys.stderr)
    gc.callbacks.append(_cb)
def _cpu_stat():   # CFS bandwidth control of the container: periods in which the cgroup ran out of its CPU quota and every thread was stopped
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            d = dict(l.split() for l in open(f).read().splitlines())
            return {k: int(v) for k, v in d.items()}
        except OSError:
            pass
    return {}


for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f"{f}: {open(f).read().strip()}")
    except OSError:
        pass
print(f"cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, torch threads {torch.get_num_threads()}, OMP_NUM_THREADS {os.environ.get('OMP_NUM_THREADS')}")
print("cpu.stat at start:", _cpu_stat())
SAMPLE = os.environ.get("SAMPLE") == "1"   # a sampling thread: where the main thread is every 2 ms, and when the sampler itself could not run
if SAMPLE:
    import threading, traceback, collections
    _samples, _stop = [], threading.Event()
    _main = threading.main_thread().ident

    def _where(fr, depth=4):
        out = []
        while fr is not None and len(out) < depth:
            out.append(f"{os.path.basename(fr.f_code.co_filename)}:{fr.f_lineno} {fr.f_code.co_name}")
            fr = fr.f_back
        return " < ".join(out)

    def _sampler():
        while not _stop.is_set():
            fr = sys._current_frames().get(_main)
            _samples.append((time.perf_counter(), _where(fr) if fr is not None else "?"))
            time.sleep(0.002)
    threading.Thread(target=_sampler, daemon=True).start()

    def _report(t0, t1):
        ss = [x for x in _samples if t0 <= x[0] <= t1]
        print(f"   [sampler] {len(ss)} samples in {1e3 * (t1 - t0):.0f} ms")
        for (ta, wa), (tb, wb) in zip(ss, ss[1:]):
            if tb - ta > 0.02:
                print(f"   [sampler] could not run for {1e3 * (tb - ta):6.1f} ms at {1e3 * (ta - t0):6.1f} ms; main thread before: {wa}")
                print(f"   [sampler]                                          after : {wb}")
        acc = collections.Counter()
        for (ta, wa), (tb, wb) in zip(ss, ss[1:]):
            acc[wa] += tb - ta
        for w, t in acc.most_common(12):
            print(f"   [sampler] {1e3 * t:6.1f} ms  {w}")
prob = synth.make_problem(2048, 512, 512, 5, N=500.0, seed=0)
X = synth.sample_torch(prob, torch.device("cuda", 0), seed=1000).t().contiguous().cpu().numpy().astype(np.float32)
for rep in range(6):
    est = SmoothNMF(n_components=5, lambda_L=1.0, simplex_H=True, simplex_W=False, shape_2d=(512, 512), max_iter=200, tol=0,
                    no_stop_criterion=True, verbose=0, random_state=0)
    torch.cuda.synchronize()
    cs0 = _cpu_stat()
    t0 = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        est.fit_transform(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        cs1 = _cpu_stat()
        print(f"rep {rep}: fit_transform {dt:.3f} s" + "".join(f"  {k} +{cs1[k] - cs0[k]}" for k in cs1 if "throttled" in k or k == "usage_usec"))
        print("   " + "\n   ".join(l for l in buf.getvalue().splitlines() if l.startswith("[fit timing")))
        if SAMPLE:
            _report(t0, t0 + dt)
