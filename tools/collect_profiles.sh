#!/bin/bash
# Everything profiles/rN_* is made from, in one run on the GPU box (from the repo root):
#   tools/collect_profiles.sh gpurun_out/r3final
# bench lines, rocprofv3 kernel traces (per-kernel summaries via tools/trace_csv_stats.py), stand-alone kernel benches.
# (The PMC traffic passes are separate: tools/pmc_traffic.sh.)
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
(time python -m pytest tests -m gpu -q) > $out/pytest.log 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --workload c4 --no-extras --no-cpu-baseline > $out/bench_c4.json 2> $out/bench_c4.err
trace() {   # name, command...
    local name=$1; shift
    mkdir -p $out/$name
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/$name/trace -o t -- "$@" > $out/$name/log.txt 2>&1
    python tools/trace_csv_stats.py $out/$name > $out/${name}_kernel_stats.txt
    rm -rf $out/$name/trace                       # (the raw traces are large; the summaries are what is kept)
}
trace bench_default python bench.py --no-cpu-baseline
trace bench_c3 python bench.py --no-extras --no-cpu-baseline
trace splat_stage python tools/splat_stage.py 3
trace splat_stage_v1 python tools/splat_stage.py 3 v1
trace c2 python tools/c2_bench.py c2
python tools/kbench.py > $out/kbench.txt 2>&1
python tools/dropin_bench.py > $out/dropin_bench.txt 2>&1
python tools/c2_bench.py > $out/c2_bench.txt 2>&1
python tools/frontend_bench.py > $out/frontend_bench.txt 2>&1
python tools/bwdbench.py > $out/bwdbench.txt 2>&1
tail -3 $out/pytest.log; cat $out/smoke.log | tail -1; cut -c1-200 $out/bench_default.json
