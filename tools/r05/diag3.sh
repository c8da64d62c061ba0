#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag3
mkdir -p $O
( time timeout 900 python tools/exp/chain_hunt3.py 1200 ) > $O/hunt3.txt 2>&1
tail -12 $O/hunt3.txt
for m in soft_bwd dense chain none; do
  ( time timeout 300 env MAIN=$m python tools/exp/chain_stress228.py 1500 ) > $O/stress_$m.txt 2>&1
  tail -4 $O/stress_$m.txt
done
( time timeout 900 python bench.py --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_diag3/bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d.get(k) for k in ('value','ms_per_step','value_cached_topology')}); print(d.get('roofline'))
PY
