"""Where the __amd_rocclr_copyBuffer launches of a kernel trace sit: (previous kernel, next kernel) pairs with counts and total time."""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ctx = defaultdict(lambda: [0, 0])
for i, (s, e, k) in enumerate(rows):
    if "copyBuffer" in k:
        p = rows[i - 1][2][:60] if i else ""
        n = rows[i + 1][2][:60] if i + 1 < len(rows) else ""
        c = ctx[(p, n)]
        c[0] += 1; c[1] += e - s
for (p, n), (cnt, t) in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{cnt:5d} {t/1e3:10.1f} us   after [{p}]  before [{n}]")
