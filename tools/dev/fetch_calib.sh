#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts in the tile kernels' own access pattern (tools/ubench/fetch_calib.hip).
# usage (GPU box, repo root): tools/dev/fetch_calib.sh <outdir>   -> <outdir>/fetch_calibration.txt
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u > $out/tcc_counters.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  for C in 100 33; do
    timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/c$C/p$i -o p -- tools/ubench/fetch_calib $C 3 > $out/c${C}_p$i.log 2>&1
  done
done
python - "$out" <<'PY' | tee $out/fetch_calibration.txt
import csv, glob, sys, collections
out = sys.argv[1]
print("FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip; rocprofv3 --kernel-trace --pmc, one counter group per pass)")
for C in (100, 33):
    nbytes = C * 768 * 1280 * 4
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/c{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"\n{C} planes of 768x1280 fp32 = {nbytes / 1e6:.1f} MB moved once per dispatch ({'>' if nbytes > 256 * 2**20 else '<'} 256 MiB Infinity Cache)")
    for k in ("stream16", "stream4", "gather4_rows", "store4_rows"):
        cs = acc.get(k, {})
        line = f"  {k:14s}"
        for name, scale in (("FETCH_SIZE", 1024.0), ("WRITE_SIZE", 1024.0)):
            if name in cs:
                v = sum(cs[name]) / len(cs[name]) * scale
                line += f" {name} {v / 1e6:8.1f} MB = {v / nbytes:5.3f} x bytes |"
        for name in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
            if name in cs:
                v = sum(cs[name]) / len(cs[name])
                line += f" {name} {v:.3e} ({nbytes / max(v, 1):.1f} B/req) |"
        print(line)
PY
