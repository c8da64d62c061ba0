#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag10
mkdir -p $O
export DC_HOP_CHAIN_GCN_MIN_NODES=0
( time timeout 600 env HUNT_CUMASK=same python tools/exp/chain_hunt_cumask.py 2000 ) > $O/hunt_cumask_same.txt 2>&1
tail -n 5 $O/hunt_cumask_same.txt
( time timeout 600 env HUNT_CUMASK=disjoint python tools/exp/chain_hunt_cumask.py 2000 ) > $O/hunt_cumask_disjoint.txt 2>&1
tail -n 5 $O/hunt_cumask_disjoint.txt
