#!/bin/bash
# The unchanged step under different GPU_MAX_HW_QUEUES (ROCm runtime: hardware queues the process' streams map onto; default 4)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_hwq; mkdir -p $O
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
for r in 1 2; do
for q in default 1 2 3 4 5 6 8; do
  if [ $q = default ]; then e="DC_NOOP=1"; else e="GPU_MAX_HW_QUEUES=$q"; fi
  v=$(env $e python bench.py $FLAGS 2>/dev/null | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"], b["serial_branches"]["value"] if isinstance(b.get("serial_branches"), dict) else "")')
  echo "GPU_MAX_HW_QUEUES=$q  $v"
done
done | tee $O/sweep.txt
