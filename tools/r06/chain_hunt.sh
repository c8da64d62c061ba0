#!/bin/bash
# round 6, VERDICT r05 item 7: the three discriminating runs + control in the 35/2,000 harness (shared CU masks, stock
# attention, tight LDS request).  usage: chain_hunt.sh [reps]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_chain_hunt
mkdir -p $O
R=${1:-2000}
export HUNT_CUMASK=same DC_FUSED_ATTN=0
for v in tight poison regstage readback; do
  ( time timeout 600 python tools/exp/run_with_lib.py tools/r06/lib_r06_$v.so tools/exp/chain_hunt_cumask.py $R ) > $O/hunt_$v.txt 2>&1
  echo "== $v: $(grep -c 'bucket differs' $O/hunt_$v.txt) differing; $(tail -n 5 $O/hunt_$v.txt | grep HUNT_CUMASK)"
done
