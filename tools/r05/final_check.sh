#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_final_check
mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -n 1
( time python bench.py ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_final_check/bench.json') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_windows']['median'], d['roofline']['frac'], d['roofline']['frac_of_moved_bytes'], d['roofline_mfma']['frac'], d['cpu_baseline']['value'])
PY
grep real $O/bench.err
