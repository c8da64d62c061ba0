#!/bin/bash
# A/B of the headline on ONE box: alternates `env A` / `env B` runs of the lean bench (headline only), 3 rounds each.
# usage: tools/exp/ab_headline.sh "DC_X=0" "DC_X=1" [steps]
A="$1"; B="$2"; STEPS="${3:-200}"
FLAGS="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-pmc --no-merged --no-backbones --no-radius100k --steps $STEPS --warmup 20"
for r in 1 2 3; do
  for e in "$A" "$B"; do
    v=$(env $e python bench.py $FLAGS 2>/dev/null | python -c 'import json,sys; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"], b["value_cached_topology"])')
    echo "round $r  [$e]  $v"
  done
done
