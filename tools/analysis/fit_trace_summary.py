#!/usr/bin/env python3
"""Summary of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run of fit_phases.py: the longest HIP API calls, the
memory copies, and the idle gaps of the device (no kernel, no copy) inside the last fit - where a whole fit's time goes that
its kernels do not explain.    python fit_trace_summary.py <dir with *_hip_api_trace.csv, *_kernel_trace.csv, *_memory_copy_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]


def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


api, ker, cpy = rows("*hip_api_trace.csv"), rows("*kernel_trace.csv"), rows("*memory_copy_trace.csv")
print(f"{len(api)} HIP API calls, {len(ker)} kernels, {len(cpy)} copies")
if not ker:
    sys.exit(0)
t_end = max(int(r["End_Timestamp"]) for r in ker)
t0 = t_end - int(0.45e9)          # the last fit (fits take 0.3-0.4 s; the run ends right after it)
f = lambda r: int(r["Start_Timestamp"]) >= t0
print("longest HIP API calls inside the last 0.45 s:")
for r in sorted([r for r in api if f(r)], key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:25]:
    print(f"   {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.2f} ms  at {(int(r['Start_Timestamp']) - t0) / 1e6:8.1f} ms  {r.get('Function', r.get('Name', '?'))}  (thread {r.get('Thread_Id', '?')})")
print("copies inside the last 0.45 s (> 1 MB):")
for r in [r for r in cpy if f(r)]:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    nbytes = int(r.get("Bytes", r.get("Size", 0)) or 0)
    if nbytes > 1 << 20:
        print(f"   {dur:9.2f} ms  at {(int(r['Start_Timestamp']) - t0) / 1e6:8.1f} ms  {nbytes / 1e6:9.1f} MB  {r.get('Direction', r.get('Name', ''))}  {nbytes / dur / 1e6 if dur else 0:7.1f} GB/s")
busy = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ker + cpy if f(r)])
gaps, cur_end = [], busy[0][1] if busy else 0
for s, e in busy[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end))
    cur_end = max(cur_end, e)
tot = sum(g for g, _ in gaps)
print(f"device busy {(sum(e - s for s, e in busy)) / 1e6:.1f} ms (overlaps counted twice), idle gaps {tot / 1e6:.1f} ms; the longest:")
for g, at in sorted(gaps, reverse=True)[:15]:
    print(f"   {g / 1e6:9.2f} ms idle at {(at - t0) / 1e6:8.1f} ms")
