#!/bin/bash
# kernel-trace timelines of the headline step with the default side stream and with a high-priority one (the 0.65 -> 0.87 ms cliff)
set -u
R=$(pwd)
O=gpurun_out/r05_prio_trace
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
Q="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones --windows 0 --steps 100 --warmup 10"
for m in off rigid_high; do
  PRIO=$m timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace_$m" -- python3 "$R/tools/r05/prio_ab.py" $Q > "$R/$O/trace_$m.log" 2>&1
  f=$(ls $R/$O/trace_$m/*/*_kernel_trace.csv 2>/dev/null | head -1)
  (cd "$R" && python tools/r05/timeline.py $f longest > "$O/timeline_$m.txt" 2>&1)
  tail -c 300 "$R/$O/trace_$m.log" | grep -o '"ms_per_step": [0-9.]*'
done
find "$R/$O" -name "*kernel_trace.csv" -delete
cat "$R/$O/timeline_off.txt" | head -60
echo ======
cat "$R/$O/timeline_rigid_high.txt" | head -70
