#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean counter value per dispatch."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append((row["Dispatch_Id"], float(row["Counter_Value"])))
    print("#", path)
    for k, cs in acc.items():
        if "free_stage" not in k:
            continue
        for c, vals in cs.items():
            per = collections.defaultdict(float)
            for d, v in vals:
                per[d] += v
            xs = sorted(per.values())
            print(f"{k[:40]:40s} {c:28s} n={len(xs):3d} max={xs[-1]:.6g} median={xs[len(xs)//2]:.6g}")
