#!/usr/bin/env python3
"""Device-side timeline of the MU loop from a rocprofv3 --kernel-trace csv: per-kernel durations, the gaps between
consecutive kernels (start of the next minus end of the previous) and the iteration period, over the last `--iters`
iterations of the run (the timed region of bench.py).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --no-cpu --steps 20 --warmup 5
    python tools/analysis/trace_gaps.py out/*/*_kernel_trace.csv --iters 20
"""
import argparse
import csv
import glob
import re
import statistics as st


def short(name):
    m = re.search(r"espm::(\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--anchor", default="h_step_ell_kernel|mu_fused|h_step_kernel", help="regex of the kernel that starts an iteration")
    args = ap.parse_args()
    for pat in args.paths:
        for path in glob.glob(pat):
            rows = list(csv.DictReader(open(path)))
            ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
            anchors = [i for i, k in enumerate(ks) if re.search(args.anchor, k[2])]
            # an iteration = an anchor kernel with a slab reduction between it and the next anchor (bench.py's per-kernel
            # timing launches at the end of the run have none)
            its = [a for a, b in zip(anchors[:-1], anchors[1:]) if any(re.search(r"w_reduce|w_finish", ks[j][2]) for j in range(a, b))]
            if len(its) < args.iters + 1:
                print(path, ": only", len(its), "iterations")
                continue
            # the last `iters` full iterations that are followed by another one (so the period is defined)
            first, last = its[-args.iters - 1], its[-1]
            seg = ks[first:last + 1]
            period = (seg[-1][0] - seg[0][0]) / args.iters / 1e3
            dur, gap = {}, {}
            for a, b in zip(seg[:-1], seg[1:]):
                dur.setdefault(short(a[2]), []).append((a[1] - a[0]) / 1e3)
                gap.setdefault(short(a[2]) + " -> " + short(b[2]), []).append((b[0] - a[1]) / 1e3)
            print(f"{path}\n  iteration period over the last {args.iters} iterations: {period:.2f} us")
            tot_d = tot_g = 0.0
            for k, v in dur.items():
                per_it = sum(v) / args.iters
                tot_d += per_it
                print(f"  kernel {k:32s} n={len(v):4d}  mean {st.mean(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f}  per iteration {per_it:8.2f} us")
            for k, v in gap.items():
                per_it = sum(v) / args.iters
                tot_g += per_it
                print(f"  gap    {k:56s} n={len(v):4d}  mean {st.mean(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f}  per iteration {per_it:7.2f} us")
            print(f"  kernels {tot_d:.2f} us + gaps {tot_g:.2f} us per iteration")


if __name__ == "__main__":
    main()
