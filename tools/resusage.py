#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin)."""
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: .*?: (Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\S+)", line)
    if not m:
        m = re.search(r"(Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, dem):
    print(f"{r.get('VGPRs','?'):>4} vgpr {r.get('SGPRs','?'):>4} sgpr scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?')}  {d[:110]}")
