#!/bin/bash
# A/B of two builds of the library on the one-flow operator at 65 x 768 x 1280 (rows front end): graph-replayed call times and per-kernel durations (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A=${1:-var_base}; B=${2:-libslrsplat}
for rep in 1 2; do for v in $A $B; do echo "== $v"; SLR_SFS_AMD_LIB=$R/slr-sfs_amd/lib/$v.so python $R/tools/dev/bin_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-40; done; done
for v in $A $B; do
  rm -rf /tmp/prof_$v
  SLR_SFS_AMD_LIB=$R/slr-sfs_amd/lib/$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v/trace -o t -- python $R/tools/dev/bin_bench.py > /tmp/prof_$v.log 2>&1 || tail -5 /tmp/prof_$v.log
  echo "== $v"; python $R/tools/trace_csv_stats.py /tmp/prof_$v 2>/dev/null | grep -i "op_rows\|rowbin" | cut -c1-150
done
