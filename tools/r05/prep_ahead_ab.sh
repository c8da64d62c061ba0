#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_prep_ahead; mkdir -p $O
( timeout 600 python -m pytest tests/test_prepare_ahead.py tests/test_full_size.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -n 4 $O/pytest.txt
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps 200 --warmup 20"
for r in 1 2 3; do
  for e in 0 1; do
    v=$(PREP_AHEAD=$e python tools/r05/prep_ahead_ab.py $FLAGS 2>$O/err_$e.txt | python -c 'import json,sys; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(b["value"], b["ms_per_step"], b["ms_per_step_windows"]["median"], b["value_cached_topology"])')
    echo "round $r  [PREP_AHEAD=$e]  $v"
  done
done | tee $O/ab.txt
tail -3 $O/err_1.txt
