#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag16
mkdir -p $O
export DC_LAUNCH_OUT=$O/launch
( time timeout 900 python tests/launch_scenarios.py dp_graphed_two_streams dp_graphed_serial --loop 25 ) > $O/dp_loop.txt 2>&1
grep -c "rc=0" $O/dp_loop.txt; tail -n 4 $O/dp_loop.txt
python - <<'PY'
import glob, json
bad = 0; n = 0
for f in sorted(glob.glob('gpurun_out/r05_diag16/launch/loop*/dp_graphed_*/dp.rank0.json')):
    d = json.load(open(f)); n += 1
    if not d.get('bit_identical'):
        bad += 1; print(f, d.get('max_abs_diff'), d.get('differing'))
print(f"{n} two-rank train-step runs, {bad} not bit-identical to the single-process mean-gradient run")
PY
