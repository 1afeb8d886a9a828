cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/copies; mkdir -p $out/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t/trace -o t -- python tools/dev/prof_copies.py > $out/t/log.txt 2>&1
python tools/trace_csv_stats.py $out/t | grep -i "copyBuffer\|fillBuffer\|calls" 
grep -v "^-\|^ *$" $out/t/log.txt | head -30 | cut -c1-220
rm -rf $out/t/trace
