#!/bin/bash
# A/B of two builds of the library on small-grid cases, per-kernel durations from rocprofv3 (run on the GPU box): tools/dev/ab_sink.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "t30" "t59 2 256 256 65 summation" "t30 1 128 240 64" "inc"; do
  for v in var_base libslrsplat; do
    rm -rf /tmp/prof_$v
    SLR_SFS_AMD_LIB=$R/slr-sfs_amd/lib/$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v/trace -o t -- python $R/tools/dev/one_case.py $c > /tmp/prof_$v.log 2>&1 || tail -5 /tmp/prof_$v.log
    echo "== $v $c"
    python $R/tools/trace_csv_stats.py /tmp/prof_$v 2>/dev/null | grep -i "slr::" | cut -c1-150
  done
done
