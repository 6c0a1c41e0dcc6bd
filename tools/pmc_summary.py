#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter over dispatches.

    python tools/pmc_summary.py gpurun_out/pmc_*/ [--kernel whitted]
"""
import csv
import glob
import sys
from collections import defaultdict

kernel_filter = "whitted"
paths = []
args = sys.argv[1:]
i = 0
while i < len(args):
    if args[i] == "--kernel":
        kernel_filter = args[i + 1]
        i += 2
    else:
        paths.append(args[i])
        i += 1
for root in paths:
    for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
        acc = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_filter in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
