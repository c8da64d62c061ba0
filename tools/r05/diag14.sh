#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag14
mkdir -p $O
( DC_HOP_CHAIN_PP=1 timeout 600 python -m pytest tests/test_hop_chain.py -x -q ) > $O/pytest_chain_pp.txt 2>&1; tail -n 8 $O/pytest_chain_pp.txt
( timeout 300 python tools/exp/hop_chain.py ) > $O/hop_chain_timing_default.txt 2>&1; tail -n 8 $O/hop_chain_timing_default.txt
( DC_HOP_CHAIN_PP=1 timeout 300 python tools/exp/hop_chain.py ) > $O/hop_chain_timing_pp.txt 2>&1; tail -n 8 $O/hop_chain_timing_pp.txt
bash tools/exp/ab_headline.sh "DC_HOP_CHAIN_PP=0" "DC_HOP_CHAIN_PP=1" 200 > $O/ab_pp.txt 2>&1; cat $O/ab_pp.txt
