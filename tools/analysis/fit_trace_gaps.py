#!/usr/bin/env python3
"""Device-side view of a whole fit: from a rocprofv3 --kernel-trace --memory-copy-trace run of fit_profile.py, the busy and idle
stretches of the GPU during the LAST fit (idle > 3 ms listed with the kernels / copies either side).
   cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/analysis/fit_profile.py
   python tools/analysis/fit_trace_gaps.py OUT"""
import csv, glob, sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "") ))
ev.sort()
# the last fit: from the last big host-to-device copy on
big = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and (e[1] - e[0]) > 20e6]
i0 = big[-1] if big else 0
t0 = ev[i0][0]
print(f"{len(ev)} device events; the last fit starts at event {i0} ({ev[i0][2]}, {(ev[i0][1] - ev[i0][0]) / 1e6:.1f} ms)")
busy = 0
end = ev[i0][0]
for s, e, name in ev[i0:]:
    if s - end > 3e6:
        print(f"  idle {(s - end) / 1e6:7.1f} ms at {(end - t0) / 1e6:7.1f} ms; next: {name}")
    if e > end:
        busy += e - max(s, end)
        end = e
print(f"span {(end - t0) / 1e6:.1f} ms, device busy {busy / 1e6:.1f} ms")
# the longest events
top = sorted(ev[i0:], key=lambda x: x[0] - x[1])[:12]
for s, e, name in top:
    print(f"  {(e - s) / 1e6:7.2f} ms at {(s - t0) / 1e6:7.1f} ms  {name}")
