#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag13
mkdir -p $O
export DC_LAUNCH_OUT=$O/launch
( time timeout 900 python tests/launch_scenarios.py bench_two_rank_gloo --loop 10 ) > $O/launch_loop_after_cap.txt 2>&1
grep -E "rc=|failing" $O/launch_loop_after_cap.txt
rm -rf $O/launch
unset DC_LAUNCH_OUT
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -n 12 $O/pytest_gpu.txt
cp gpurun_out/parity_distances.json $O/ 2>/dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
