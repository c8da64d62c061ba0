#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_rccl_probe; mkdir -p $O
for r in 1 2; do for m in none pg ar ar_tl ar_graph ar_graph_tl; do timeout 300 python tools/r05/rccl_step_probe.py $m 2>$O/err_$m.txt | grep "ms per step" || echo "$m: no result (rc $?)"; done; done | tee $O/probe.txt
grep -h -i "error\|Traceback" -A3 $O/err_*.txt | head -20
