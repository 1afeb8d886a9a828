#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace (per-kernel durations).
# usage (on the GPU box, from the repo root): tools/profile_bench.sh gpurun_out/prof_r1
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- $CMD > $out/trace.log 2>&1
# (PMC passes: tools/pmc_traffic.sh on the splat stage -- rocprofv3 --pmc segfaults with the MIOpen pipeline)
grep -h '"metric"' $out/*.log | cut -c1-300
