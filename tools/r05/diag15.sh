#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag15
mkdir -p $O
( timeout 600 python -m pytest tests/test_narrow_dense.py -x -q ) > $O/pytest_narrow.txt 2>&1; tail -n 12 $O/pytest_narrow.txt
( timeout 300 python tools/r05/narrow_dense_time.py ) > $O/narrow_dense_time.txt 2>&1; cat $O/narrow_dense_time.txt | tail -n 8
bash tools/exp/ab_headline.sh "DC_DENSE_NARROW=0" "DC_DENSE_NARROW=1" 200 > $O/ab_narrow_dense.txt 2>&1; cat $O/ab_narrow_dense.txt
