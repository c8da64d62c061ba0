#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag4
mkdir -p $O
( time timeout 700 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_poison.so tools/exp/chain_hunt.py 2000 ) > $O/hunt_poison.txt 2>&1
grep -c "bucket differs" $O/hunt_poison.txt; grep -i -c "nan" $O/hunt_poison.txt; tail -n 3 $O/hunt_poison.txt
( time timeout 700 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_lds160.so tools/exp/chain_hunt.py 2000 ) > $O/hunt_lds160.txt 2>&1
grep -c "bucket differs" $O/hunt_lds160.txt; tail -n 3 $O/hunt_lds160.txt
( time timeout 600 python -m pytest tests/test_hop_chain.py tests/test_node_order.py -x -q ) > $O/pytest_chain.txt 2>&1
tail -n 5 $O/pytest_chain.txt
