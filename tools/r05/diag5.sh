#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag5
mkdir -p $O
( time timeout 700 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_r04chain.so tools/exp/chain_hunt.py 2000 ) > $O/hunt_r04chain.txt 2>&1
grep -c "bucket differs" $O/hunt_r04chain.txt; tail -n 4 $O/hunt_r04chain.txt
( time timeout 700 env HUNT_TAPS=1 python tools/exp/chain_hunt.py 2000 ) > $O/hunt_control.txt 2>&1
grep -c "bucket differs" $O/hunt_control.txt; tail -n 4 $O/hunt_control.txt
