#!/bin/bash
# usage (GPU box): bash tools/dev/prof_variants.sh "<fe> <case>" var1 var2 ... -> per-kernel times of tools/dev/fe_one.py per library variant
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
args="$1"; shift
for v in "$@"; do
  rm -rf gpurun_out/prof_fe
  if [ "$v" = default ]; then lib=""; else lib=$PWD/slr-sfs_amd/lib/var_$v.so; fi
  SLR_SFS_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_fe/trace -o t -- python tools/dev/fe_one.py $args > gpurun_out/prof_fe.log 2>&1
  echo "=== $v ($args)"
  python tools/trace_csv_stats.py gpurun_out/prof_fe | grep -E "slr::" | grep -v euler | cut -c1-140
done
