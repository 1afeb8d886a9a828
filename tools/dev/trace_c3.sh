cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/skipfuse; mkdir -p $out/bench_c3
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/bench_c3/trace -o t -- python bench.py --no-extras --no-cpu-baseline > $out/bench_c3/log.txt 2>&1
python tools/trace_csv_stats.py $out/bench_c3 > $out/bench_c3_kernel_stats.txt
python tools/dev/copy_context.py $out/bench_c3 > $out/copy_context.txt
rm -rf $out/bench_c3/trace
head -24 $out/bench_c3_kernel_stats.txt | cut -c1-150
