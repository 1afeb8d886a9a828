#!/bin/bash
# usage (on the GPU box): bash tools/dev/prof_fe.sh <front end 0|1|2> <case...>   -> per-kernel times of that front end
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
fe=$1; shift
for c in "$@"; do
  rm -rf gpurun_out/prof_fe
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_fe/trace -o t -- python tools/dev/fe_one.py $fe $c > gpurun_out/prof_fe.log 2>&1
  echo "=== front end $fe, $c"
  python tools/trace_csv_stats.py gpurun_out/prof_fe | grep -E "slr::|kernel " | cut -c1-140
done
