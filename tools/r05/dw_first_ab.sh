#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_dw_first; mkdir -p $O
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
for r in 1 2 3 4 5; do
  for m in ${ARMS:-last,last first,last mid,last first,mid mid,first last,mid}; do
    v=$(DWF=$m python tools/r05/dw_first_ab.py $FLAGS 2>$O/err.txt | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"])')
    echo "round $r  [DWF=$m]  $v"
  done
done | tee $O/ab.txt
