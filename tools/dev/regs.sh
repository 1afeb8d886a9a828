#!/bin/bash
# usage: tools/dev/regs.sh <file.hip> [extra -D flags...]  -- registers / scratch / code size of every kernel of a source file (gfx950 device pass)
f=$1; shift
cd "$(dirname "$0")/../../slr-sfs_amd/csrc"
out=/tmp/isa/$(basename $f .hip)_regs.s; mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" --cuda-device-only -S $f -o $out 2>&1 | grep -v "recognized\|hip-link"
awk '/^_Z[A-Za-z0-9_]*:/ {name=$1} /; codeLenInByte/ {c=$4} /; TotalNumSgprs/ {s=$3} /; NumVgprs:/ {v=$3} /; ScratchSize/ {sc=$3} /; Occupancy/ {printf "%-90s code %6d sgpr %3d vgpr %3d scratch %4d occ %d\n", substr(name,1,90), c, s, v, sc, $3}' $out | c++filt | cut -c1-170
