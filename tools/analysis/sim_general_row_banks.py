#!/usr/bin/env python3
"""What ordering the GENERAL rows of the sparse store by LDS bank would buy (third part): a model of the table
gathers of one list group - 64 lanes, read in groups of 8, a group taking as many passes as its fullest bank quad has distinct
rows - on bench-like spectra, for the entries with a count of 2 or more in channel order (the builder today), in random order,
sorted by the lane-rotated quad, and packed into slots with holes like the unit rows."""
import numpy as np
rng = np.random.default_rng(0)
n, lanes, groups = 2048, 64, 40
# per-pixel spectra like the bench: 500 counts per pixel over 2048 channels, smooth-ish intensity
chan = np.arange(n)
base = np.exp(-((chan - 600) / 400.0) ** 2) + 0.6 * np.exp(-((chan - 1400) / 250.0) ** 2) + 0.15
base /= base.sum()
def cost_rows(lists, group_of=8):
    # lists: list over lanes of arrays of channel indices (position j = row j); cost = sum over positions and lane groups of max multiplicity of bank quad (idx & 15) among DISTINCT idx
    L = max(len(a) for a in lists)
    tot = 0; rows = 0
    for j in range(L):
        for g0 in range(0, lanes, group_of):
            idx = [lists[l][j] for l in range(g0, g0 + group_of) if j < len(lists[l])]
            if not idx: continue
            idx = np.unique(idx)
            tot += np.bincount(idx & 15, minlength=16).max()
            rows += 1
    return tot / max(rows, 1)
res = {k: [] for k in ("channel order", "sorted by rotated key", "slots + holes (ideal packing)", "random")}
for g in range(groups):
    X = rng.poisson(500.0 * base * rng.uniform(0.7, 1.3), size=(lanes, n))
    gen = [np.nonzero(X[l] >= 2)[0] for l in range(lanes)]   # (overflow units ignored)
    res["channel order"].append(cost_rows(gen))
    res["random"].append(cost_rows([rng.permutation(a) for a in gen]))
    srt = []
    for l, a in enumerate(gen):
        key = (a - l) & 15
        srt.append(a[np.argsort(key, kind="stable")])
    res["sorted by rotated key"].append(cost_rows(srt))
    # ideal: position p = ((q - l) & 15) + 16 r for in-bucket, overflow into holes (greedy), remainder in the tail
    ideal = []
    for l, a in enumerate(gen):
        q = a & 15
        ng = len(a); slots = ng // 16
        out = np.full(ng, -1)
        rank = np.zeros(16, int)
        overflow = []
        for c in a:
            b = c & 15
            if rank[b] < slots:
                out[((b - l) & 15) + 16 * rank[b]] = c; rank[b] += 1
            else:
                overflow.append(c)
        free = [p for p in range(ng) if out[p] < 0]
        for p, c in zip(free, overflow):
            out[p] = c
        ideal.append(out)
    res["slots + holes (ideal packing)"].append(cost_rows(ideal))
for k, v in res.items():
    print(f"{k:32s}: {np.mean(v):.3f} LDS passes per 8-lane read group (1.0 = conflict-free)")
print("general entries per lane:", np.mean([len(a) for a in gen]))
