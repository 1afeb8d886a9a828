#!/bin/bash
# HBM traffic of the splat kernels: FETCH_SIZE and WRITE_SIZE in separate passes (kernel-trace only).
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o t -- python tools/splat_stage.py > $out/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o t -- python tools/splat_stage.py > $out/write.log 2>&1
tail -1 $out/fetch.log; tail -1 $out/write.log
# then: python tools/pmc_traffic.py $out profiles/rN_splat_traffic.json
