#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name (espm:: kernels only)."""
import csv, glob, sys, collections
for pat in sys.argv[1:]:
    for f in glob.glob(pat):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "espm::" not in name:
                continue
            acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, cs in acc.items():
            print(name)
            for c, v in cs.items():
                print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
