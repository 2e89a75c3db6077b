#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, the counter value per dispatch
(summed over the dispatch's dimensions/instances): number of dispatches, median and max.

    python tools/pmc_summary.py [--kernel SUBSTR] file.csv ...
"""
import collections
import csv
import sys

args = sys.argv[1:]
want = "stage_kernel"
if args and args[0] == "--kernel":
    want = args[1]
    args = args[2:]
for path in args:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append((row["Dispatch_Id"], float(row["Counter_Value"])))
    print("#", path)
    for k, cs in acc.items():
        if want not in k:
            continue
        for c, vals in sorted(cs.items()):
            per = collections.defaultdict(float)
            for d, v in vals:
                per[d] += v
            xs = sorted(per.values())
            print(f"{k[:44]:44s} {c:28s} n={len(xs):3d} max={xs[-1]:.6g} median={xs[len(xs)//2]:.6g}")
