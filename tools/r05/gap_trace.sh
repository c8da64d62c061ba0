#!/bin/bash
# kernel-trace timelines of the headline step: default and --serial-branches (where does the 40 us gap in front of the soft branch's
# k_pack_weights come from?)
set -u
R=$(pwd)
O=gpurun_out/r05_gap_trace
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
Q="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones --windows 0 --steps 100 --warmup 10"
for m in ${MODES:-default serial}; do
  X=""; [ $m = serial ] && X="--serial-branches"
  timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace_$m" -- python3 "$R/bench.py" $Q $X > "$R/$O/trace_$m.log" 2>&1
  f=$(ls $R/$O/trace_$m/*/*_kernel_trace.csv 2>/dev/null | head -1)
  (cd "$R" && python tools/r05/timeline.py $f longest > "$O/timeline_$m.txt" 2>&1)
done
find "$R/$O" -name "*kernel_trace.csv" -delete
head -30 "$R/$O/timeline_default.txt"
echo ======
head -30 "$R/$O/timeline_serial.txt"
