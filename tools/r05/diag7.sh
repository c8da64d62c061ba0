#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag7
mkdir -p $O
# (diag4-6 ran the variant libraries WITHOUT the LDS-table form on small graphs: run_with_lib.py imports the package before
# chain_hunt.py sets DC_HOP_CHAIN_GCN_MIN_NODES - the variable has to come from the shell)
export DC_HOP_CHAIN_GCN_MIN_NODES=0
( time timeout 900 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_poisonend.so tools/exp/chain_hunt.py 3000 ) > $O/hunt_poisonend.txt 2>&1
grep -c "bucket differs" $O/hunt_poisonend.txt; grep -c -i "nan" $O/hunt_poisonend.txt; tail -n 4 $O/hunt_poisonend.txt
( time timeout 700 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_lds160.so tools/exp/chain_hunt.py 2000 ) > $O/hunt_lds160.txt 2>&1
grep -c "bucket differs" $O/hunt_lds160.txt; tail -n 4 $O/hunt_lds160.txt
