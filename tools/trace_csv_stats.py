#!/usr/bin/env python
"""Per-kernel duration summary of a rocprofv3 kernel_trace.csv (what --stats prints) + mean PMC values."""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
d = defaultdict(list)
for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values()) or 1
print(f"{'kernel':78s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{k[:78]:78s} {len(v):6d} {sum(v)/len(v)/1e3:10.1f} {min(v)/1e3:10.1f} {max(v)/1e3:10.1f} {sum(v)/1e6:10.3f} {100*sum(v)/tot:6.1f}")
for name in ("fetch", "write"):
    acc = defaultdict(list)
    for f in glob.glob(root + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "slr::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"PMC {c:12s} {k:70s} n={len(v):4d} mean={sum(v)/len(v):14.1f} (KB per dispatch)")
