#!/bin/bash
# usage (on the GPU box, from the repo root): tools/dev/r5_exp.sh <outdir> <lib1> <lib2> ...   ("default" = the product library)
# per library variant: the timed C3 stage line (tile kernel / stage), the one-flow operator (both front ends, small and full grids),
# after a short parity run on the product library.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
if [ -n "$PARITY" ]; then
  (time python -m pytest tests -m gpu -x -q -k "$PARITY") > $out/parity.log 2>&1; tail -3 $out/parity.log
fi
for v in "$@"; do
  if [ "$v" = default ]; then unset SLR_SFS_AMD_LIB; else export SLR_SFS_AMD_LIB=$PWD/slr-sfs_amd/lib/var_$v.so; fi
  echo "=== $v" | tee -a $out/lines.txt
  python tools/dev/stage_line.py 2>&1 | grep -v amdgpu | tee -a $out/lines.txt
  [ -n "$NOFE" ] || python tools/frontend_bench.py 2>&1 | grep -v amdgpu | tee -a $out/lines.txt
done
