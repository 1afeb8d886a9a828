#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "splat_tile"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if pat in k:
            acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
