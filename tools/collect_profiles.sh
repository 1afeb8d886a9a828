#!/bin/bash
# Everything profiles/rN_* is made from, in one run on the GPU box (from the repo root):
#   tools/collect_profiles.sh gpurun_out/r6final
# bench lines, rocprofv3 kernel traces (per-kernel summaries via tools/trace_csv_stats.py), PMC passes (SQ / TCC counters, FETCH_SIZE and
# WRITE_SIZE in separate runs, kernel-trace only) for the fused clip kernel (C3 and C4), the one-flow operator (rows at 768x1280, scan at
# config C2) and the backward kernel, stand-alone kernel benches.
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
trace c2_smooth_t30 python tools/dev/fe_shape.py 1 64 256 480 softmax 30
trace train_t30 python tools/dev/fe_shape.py 2 65 256 256 summation 30
trace train_t59 python tools/dev/fe_shape.py 2 65 256 256 summation 59
trace frontends python tools/dev/fe_one.py 2
trace bwd python tools/bwdbench.py
pmc() {     # name, kernel pattern, units of work per dispatch, command...
    local name=$1 pat=$2 units=$3; shift 3
    bash tools/pmc_run.sh $out/pmc_$name "$@"
    python tools/pmc_summary.py $out/pmc_$name "$pat" > $out/pmc_${name}_counters.txt
    python tools/pmc_kernel_traffic.py $out/pmc_$name "$pat" $out/traffic_$name.json $units
    find $out/pmc_$name -name "*.csv" -delete; find $out/pmc_$name -name "*.log" -delete
}
pmc clip_c3 "clip_tile_kernel<false, false, true>" 15 python tools/splat_stage.py
pmc clip_c4 "clip_tile_kernel<true, false, true>" 15 python tools/splat_stage.py v1
pmc op_rows_t30 "op_rows_kernel<false, false, false>" 1 python tools/dev/fe_one.py 2 t30
pmc op_rows_t59 "op_rows_kernel<false, false, false>" 1 python tools/dev/fe_one.py 2 t59
pmc op_scan_c2 "op_scan_kernel<true, false>" 1 python tools/dev/fe_one.py 1 c2
pmc grad_t30 "grad_tile_kernel<true, true>" 1 python tools/bwdbench.py t30
python tools/kbench.py > $out/kbench.txt 2>&1
python tools/frontend_bench.py > $out/frontend_bench.txt 2>&1
python tools/small_grid_bench.py > $out/small_grid_bench.txt 2>&1
python tools/dev/fuzz_frontends.py 3000 > $out/fuzz.txt 2>&1; python tools/dev/fuzz_clip.py >> $out/fuzz.txt 2>&1; python tools/dev/fuzz_backward.py >> $out/fuzz.txt 2>&1; python tools/dev/soak_frontends.py >> $out/fuzz.txt 2>&1
python tools/bwdbench.py > $out/bwdbench.txt 2>&1
python tools/dev/convbench_f32.py > $out/convbench_f32.txt 2>&1
tail -3 $out/pytest.log; cat $out/smoke.log | tail -2; cut -c1-200 $out/bench_default.json
