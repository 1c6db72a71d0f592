#!/usr/bin/env python3
"""Print the espm:: rows of a rocprofv3 kernel_stats.csv (name, calls, avg us, min us, max us)."""
import csv, glob, sys
for path in sys.argv[1:]:
    for f in glob.glob(path):
        for r in list(csv.reader(open(f)))[1:]:
            if "espm::" in r[0]:
                print(f"{r[0][:72]:72s} {int(r[1]):5d} {float(r[3])/1e3:9.1f} {float(r[5])/1e3:9.1f} {float(r[6])/1e3:9.1f}")
