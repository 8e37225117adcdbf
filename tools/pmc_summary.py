#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean of each counter per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
