#!/usr/bin/env python3
"""Device time line of the LAST fit of a rocprofv3 --kernel-trace --memory-copy-trace run of fit_timing.py: every idle gap of the
device above 2 ms with the kernel (or copy) that ended before it and the one that started after it, and between the gaps the busy
time of the stretch and its most frequent kernel - which section of fit_transform the device waits in, and for how long.
    python fit_timeline.py <dir with *_kernel_trace.csv, *_memory_copy_trace.csv> [window seconds, default 0.5]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5


def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows("*kernel_trace.csv")]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %.1f MB" % (r.get("Direction", ""), int(r.get("Bytes", r.get("Size", 0)) or 0) / 1e6))
       for r in rows("*memory_copy_trace.csv")]
ev.sort()
t_end = max(e for _, e, _ in ev)
t0 = t_end - int(win * 1e9)
ev = [x for x in ev if x[0] >= t0]
print(f"{len(ev)} kernels and copies in the last {win} s")
seg_start, cur_end, busy, names, last = ev[0][0], ev[0][1], 0, collections.Counter(), ev[0][2]
for s, e, n in ev:
    if s - cur_end > 2e6:
        top = ", ".join(f"{k} x{v}" for k, v in names.most_common(3))
        print(f"{(seg_start - t0) / 1e6:8.1f} .. {(cur_end - t0) / 1e6:8.1f} ms  busy {busy / 1e6:7.1f} ms of {(cur_end - seg_start) / 1e6:7.1f}   [{top}]")
        print(f"      idle {(s - cur_end) / 1e6:7.1f} ms   after: {last}   before: {n}")
        seg_start, busy, names = s, 0, collections.Counter()
    busy += e - s
    names[n] += 1
    if e >= cur_end:
        cur_end, last = e, n
top = ", ".join(f"{k} x{v}" for k, v in names.most_common(3))
print(f"{(seg_start - t0) / 1e6:8.1f} .. {(cur_end - t0) / 1e6:8.1f} ms  busy {busy / 1e6:7.1f} ms of {(cur_end - seg_start) / 1e6:7.1f}   [{top}]")
