#!/bin/bash
# round 5: what profiles/r05/ cites from the final code, one pass on a GPU box (trimmed tools/refresh_profiles.sh)
set -u
R=$(pwd)
O=gpurun_out/r05_final
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
(cd "$R" && S=$(date +%s) && timeout -s KILL 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench wall $(( $(date +%s) - S )) s" | tee "$O/bench_wall.txt"; tail -c 200 "$O/bench.json")
Q="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones --windows 0"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_default" -- python3 "$R/bench.py" $Q > "$R/$O/stats_default.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_serial" -- python3 "$R/bench.py" $Q --serial-branches > "$R/$O/stats_serial.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/pmc_a" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_a.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_b" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_b.log" 2>&1
(cd "$R" && python tools/pmc_dense.py --parse "$O/pmc_a" "$O/pmc_b" > "$O/dense_pmc.json" 2> "$O/dense_pmc.err")
(cd "$R" && timeout -s KILL 300 python tools/exp/hop_chain.py > "$O/hop_chain_timing.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/kbench.py > "$O/kbench.txt" 2>&1)
f=$(ls $R/$O/stats_default/*/*_kernel_trace.csv 2>/dev/null | head -1)
(cd "$R" && python tools/r05/timeline.py $f > "$O/timeline_default.txt" 2>&1)
find "$R/$O" -name "*kernel_trace.csv" -delete
find "$R/$O" -name "*counter_collection.csv" -delete
ls "$R/$O"
