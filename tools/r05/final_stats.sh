#!/bin/bash
# rocprofv3 --kernel-trace --stats of the lean bench of the final code, two streams and one (profiles/r05/z_*)
set -u
R=$(pwd)
O=gpurun_out/r05_final_stats
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
Q="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones --windows 0"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_default" -- python3 "$R/bench.py" $Q > "$R/$O/stats_default.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_serial" -- python3 "$R/bench.py" $Q --serial-branches > "$R/$O/stats_serial.log" 2>&1
for m in default serial; do
  f=$(ls $R/$O/stats_$m/*/*_kernel_stats.csv 2>/dev/null | head -1); cp "$f" "$R/$O/${m}_kernel_stats.csv"
  tail -c 400 "$R/$O/stats_$m.log" | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*' | head -2
done
find "$R/$O" -name "*kernel_trace.csv" -delete
head -8 "$R/$O/serial_kernel_stats.csv" | cut -c1-120
