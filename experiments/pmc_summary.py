#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "ndt2d" not in k:
            continue
        print(k)
        for c, vals in sorted(v.items()):
            print("   %-26s n=%d avg=%.5g" % (c, len(vals), sum(vals) / len(vals)))
