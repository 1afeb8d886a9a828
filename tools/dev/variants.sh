#!/bin/bash
# usage: tools/dev/variants.sh "<python command>" var1 var2 ...   (runs the command with SLR_SFS_AMD_LIB = each variant; "default" = the product library)
cmd="$1"; shift
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = default ]; then $cmd; else SLR_SFS_AMD_LIB=$PWD/slr-sfs_amd/lib/var_$v.so $cmd; fi
done
