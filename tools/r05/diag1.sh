#!/bin/bash
# round 5, first GPU call: (1) loop the two-rank gloo launch with per-rank logs + watchdog to catch the r04 hang,
# (2) the chain-launch difference hunt with element-level taps, (3) the back-to-back launch comparison
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag1
mkdir -p $O
export DC_LAUNCH_OUT=$O/launch
( time timeout 900 python tests/launch_scenarios.py bench_two_rank_gloo --loop 12 ) > $O/launch_loop.txt 2>&1
( time timeout 600 env HUNT_TAPS=1 python tools/exp/chain_hunt.py 500 ) > $O/hunt_taps.txt 2>&1
( time timeout 600 env HUNT_DOUBLE=1 python tools/exp/chain_hunt.py 500 ) > $O/hunt_double.txt 2>&1
tail -5 $O/launch_loop.txt $O/hunt_taps.txt $O/hunt_double.txt
# keep the pulled directory small: only failing loops' logs
for d in $O/launch/loop*; do
  if ! grep -q '"rc": [1-9]' $d/results.json 2>/dev/null; then rm -rf $d; fi
done
