#!/bin/bash
# Clock and power of the GPU while the 128->128 3x3 convolution runs back to back (evidence for the
# "clock(power)-limited" statement of DESIGN.md 3.4): rocm-smi sampled once a second next to tools/dev/convloop.py.
out=$1; mode=$2          # mode: empty = split-f16 rung, fp32 = the fp32 rung
mkdir -p $out
python tools/dev/convloop.py 14 $mode > $out/convloop.log 2>&1 &
pid=$!
sleep 6                                   # import + warm-up
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)|hotspot" | tr -s ' ' | sed "s/^/t=$i /" >> $out/smi.log
  sleep 1
done
wait $pid
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | sed "s/^/idle /" >> $out/smi.log
cat $out/convloop.log | tail -3
cat $out/smi.log
