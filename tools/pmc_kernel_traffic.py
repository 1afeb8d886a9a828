#!/usr/bin/env python
"""HBM-side traffic of one kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_run.sh (p3 / p4), per dispatch:
    python tools/pmc_kernel_traffic.py <pmc dir> "<kernel name pattern>" <out.json> [units of work per dispatch, e.g. 15 frames]
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE counts 128-byte requests as 64 bytes -> x2; WRITE_SIZE exact; both
in KiB.  The x2 was re-calibrated in round 5 on the tile kernels' own access pattern (4-byte buffer_load gathers with the plane offset in
an SGPR, tools/ubench/fetch_calib.hip -> profiles/r5_fetch_calibration.txt: FETCH_SIZE = 0.500 x bytes, every fabric read a 128-byte
request).  Cross-check without any factor, from the request-size counters of passes p6 / p7 (tools/pmc_run.sh):
read bytes = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (`read_MB_by_request_size`).  The JSON carries the hash of the kernel sources."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402
src, pat, dst = sys.argv[1], sys.argv[2], sys.argv[3]
units = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0


def mean_counter(sub, name):
    vals = []
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row.get("Kernel_Name", "") and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch, n1 = mean_counter("p3", "FETCH_SIZE")
write, n2 = mean_counter("p4", "WRITE_SIZE")
r32, _ = mean_counter("p6", "TCC_EA0_RDREQ_32B_sum")
r64, _ = mean_counter("p7", "TCC_EA0_RDREQ_64B_sum")
r128, _ = mean_counter("p7", "TCC_EA0_RDREQ_128B_sum")
rall, _ = mean_counter("p6", "TCC_EA0_RDREQ_sum")
out = {"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh), mean per dispatch of the named kernel; "
                   "FETCH_SIZE x2 on gfx950 (128-byte requests tallied at 64), WRITE_SIZE exact",
       "kernel_pattern": pat, "source_sha16": csrc_hash(), "dispatches_seen": [n1, n2],
       "read_MB_per_dispatch": None if fetch is None else round(fetch * 1024 * 2.0 / 1e6, 1),
       "written_MB_per_dispatch": None if write is None else round(write * 1024 / 1e6, 1)}
if None not in (r32, r64, r128):
    out["read_MB_by_request_size"] = round((32 * r32 + 64 * r64 + 128 * r128) / 1e6, 1)
    out["read_requests"] = {"all": rall, "32B": r32, "64B": r64, "128B": r128}
if fetch is not None and write is not None:
    out["traffic_bytes_per_dispatch"] = int(fetch * 1024 * 2.0 + write * 1024)
    out["units_per_dispatch"] = units              # (bench.py: traffic per unit of work x the units of ITS launches)
    out["traffic_bytes_per_unit"] = int((fetch * 1024 * 2.0 + write * 1024) / units)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
