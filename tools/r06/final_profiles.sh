#!/bin/bash
# round 6: what profiles/r06/ cites from the final code besides the default bench run (tools/r06/check.sh): rocprofv3 kernel
# statistics of the two-stream and the one-stream headline step, the dense blocks' PMC summary, kernel micro-benchmarks
set -u
R=$(pwd)
O=gpurun_out/r06_final
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
Q="--headline-only --settle 30 --steps 50 --warmup 5 --windows 0 --no-pmc --no-kernel-trace"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_default" -- python3 "$R/bench.py" $Q > "$R/$O/stats_default.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_serial" -- python3 "$R/bench.py" $Q --serial-branches > "$R/$O/stats_serial.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/pmc_a" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_a.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_b" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_b.log" 2>&1
(cd "$R" && python tools/pmc_dense.py --parse "$O/pmc_a" "$O/pmc_b" > "$O/dense_pmc.json" 2> "$O/dense_pmc.err")
for m in default serial; do
  f=$(ls $R/$O/stats_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$R/$O/${m}_kernel_stats.csv"
done
f=$(ls $R/$O/stats_default/*/*_kernel_trace.csv 2>/dev/null | head -1)
(cd "$R" && python tools/r05/timeline.py $f > "$O/timeline_default.txt" 2>&1)
find "$R/$O" -name "*kernel_trace.csv" -delete
find "$R/$O" -name "*counter_collection.csv" -delete
rm -rf "$R/$O/stats_default" "$R/$O/stats_serial" "$R/$O/pmc_a" "$R/$O/pmc_b"
ls "$R/$O"; head -c 1500 "$R/$O/dense_pmc.json"
