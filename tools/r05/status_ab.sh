#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_status; mkdir -p $O
( DC_SKIP_LAUNCH=1 timeout 900 python -m pytest tests/test_segmented_build.py tests/test_train_loop.py tests/test_merged.py tests/test_full_size.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
for r in 1 2 3; do
  for e in 0 1; do
    v=$(SHARED_STATUS=$e python tools/r05/status_ab.py $FLAGS 2>$O/err_$e.txt | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"])')
    echo "round $r  [SHARED_STATUS=$e]  $v"
  done
done | tee $O/ab.txt
