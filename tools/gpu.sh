#!/bin/bash
# Build everything (library, oracle, host example) and run a command on an MI355X box through gpurun.
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
exec timeout $(( $1 + 1500 )) /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
