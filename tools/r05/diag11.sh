#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag11
mkdir -p $O
export DC_HOP_CHAIN_GCN_MIN_NODES=0
( time timeout 600 env HUNT_CUMASK=same DC_FUSED_ATTN=1 python tools/exp/chain_hunt_cumask.py 2000 ) > $O/hunt_same_fusedattn.txt 2>&1
tail -n 5 $O/hunt_same_fusedattn.txt
( time timeout 600 env HUNT_CUMASK=same DC_FUSED_ATTN=0 python tools/exp/chain_hunt_cumask.py 1000 ) > $O/hunt_same_stockattn.txt 2>&1
tail -n 5 $O/hunt_same_stockattn.txt
for o in attn_stock gemm softmax dense none; do
  ( timeout 300 env OTHER=$o python tools/exp/chain_coresident.py 6000 ) > $O/coresident_$o.txt 2>&1
  tail -n 3 $O/coresident_$o.txt
done
