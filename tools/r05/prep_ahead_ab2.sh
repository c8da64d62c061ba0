#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_prep_ahead; mkdir -p $O
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
run() { # label, env...
  label=$1; shift
  v=$(env "$@" python tools/r05/prep_ahead_ab.py $FLAGS 2>$O/err_x.txt | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"])')
  echo "[$label]  $v"
}
for r in 1 2; do
  run "off" PREP_AHEAD=0
  run "on, one helper" PREP_AHEAD=1 PREP_HELPERS=1
  run "off, 8 hw queues" PREP_AHEAD=0 GPU_MAX_HW_QUEUES=8
  run "on, two helpers, 8 hw queues" PREP_AHEAD=1 GPU_MAX_HW_QUEUES=8
  run "on, one helper, 8 hw queues" PREP_AHEAD=1 PREP_HELPERS=1 GPU_MAX_HW_QUEUES=8
done | tee $O/ab2.txt
