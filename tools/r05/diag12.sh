#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag12
mkdir -p $O
# the fix: every chain workgroup owns the whole LDS.  Shared CUs, stock attention (the failing configuration):
( time timeout 900 env HUNT_CUMASK=same python tools/exp/chain_hunt_cumask.py 3000 ) > $O/hunt_fixed_3000.txt 2>&1
tail -n 4 $O/hunt_fixed_3000.txt
# control on the same box: the same library with the chain kernels asking only for the LDS their slice needs
( time timeout 600 env HUNT_CUMASK=same python tools/exp/run_with_lib.py tools/r05/lib_ldstight.so tools/exp/chain_hunt_cumask.py 1000 ) > $O/hunt_tight_1000.txt 2>&1
tail -n 4 $O/hunt_tight_1000.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -n 12 $O/pytest_gpu.txt
cp gpurun_out/parity_distances.json $O/ 2>/dev/null
