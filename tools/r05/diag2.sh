#!/bin/bash
# round 5, second GPU call: the whole GPU suite as the driver runs it (launch scenarios at session start), then the
# bracketed chain-launch hunt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag2
mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 ) > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
cp gpurun_out/parity_distances.json $O/ 2>/dev/null
( time timeout 900 python tools/exp/chain_hunt2.py 1500 ) > $O/hunt2.txt 2>&1
tail -40 $O/hunt2.txt
