#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_soft_prep; mkdir -p $O
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
for r in 1 2 3; do
  for m in 0 1; do
    v=$(SOFTPREP=$m python tools/r05/soft_prep_ab.py $FLAGS 2>$O/err_$m.txt | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"])')
    echo "round $r  [SOFTPREP=$m]  $v"
  done
done | tee $O/ab.txt
tail -3 $O/err_1.txt
