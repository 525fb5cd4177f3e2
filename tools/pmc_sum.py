#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per (kernel, counter): python tools/pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[(k[:60], r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k[:60], r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} n={cnt[(k, c)]:4d} sum={v:.4g} avg={v / cnt[(k, c)]:.4g}")
