#!/usr/bin/env python
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/rN_splat_traffic.json:
HBM-side bytes of the fused tile kernel per FRAME of work, with the hash of the kernel sources it was taken on
(bench.py labels a `traffic` figure from another build as stale).
    python tools/pmc_traffic.py gpurun_out/<dir> profiles/r3_splat_traffic.json
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE counts 128-byte requests as 64 bytes -> x2
(calibrated in round 1 on known byte counts with this access pattern); WRITE_SIZE exact; both in KiB."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import H, W, csrc_hash, splat_alg_bytes  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "clip_tile_kernel<false, false>"          # (v1: "clip_tile_kernel<true, false>")


def mean_counter(sub, name):
    vals = []
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row.get("Kernel_Name", "") and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    assert vals, (sub, name)
    return sum(vals) / len(vals), len(vals)


fetch, n1 = mean_counter("fetch", "FETCH_SIZE")
write, n2 = mean_counter("write", "WRITE_SIZE")
fpl = 60.0 / n1                                    # tools/splat_stage.py renders one 60-frame clip: 3 launches of 16 + one of 12 -> 15 per launch
read_b, write_b = fetch * 1024 * 2.0 / fpl, write * 1024 / fpl
out = {
    "_comment": "HBM-side traffic of the fused splat tile kernel: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate "
                "passes on tools/splat_stage.py (tools/pmc_traffic.sh), summarised by tools/pmc_traffic.py; values per FRAME of work "
                "(one launch = 60 / launches frames on average).  FETCH_SIZE x2 on gfx950 (128-byte requests tallied at 64), WRITE_SIZE exact.",
    "kernel": "slr::" + KERNEL + "(ClipBatch)",
    "source_sha16": csrc_hash(),
    "frames_per_launch": fpl, "launches_seen": [n1, n2],
    "FETCH_SIZE_KiB_raw_per_launch": round(fetch, 1), "WRITE_SIZE_KiB_raw_per_launch": round(write, 1), "fetch_correction": 2.0,
    "read_MB": round(read_b / 1e6, 1), "written_MB": round(write_b / 1e6, 1),
    "traffic_bytes_per_launch": int(read_b + write_b),
    "algorithmic_bytes_per_launch": splat_alg_bytes(65),
    "min_bytes_per_launch": (64 + 1 + 4 + 64) * H * W * 4,
    "_units": "traffic / algorithmic / min bytes: per frame of work (bench.py multiplies by its frames_per_launch)",
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
