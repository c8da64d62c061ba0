#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag9
mkdir -p $O
( timeout 600 python -m pytest tests/test_fused_pack.py -x -q ) > $O/pytest_fused.txt 2>&1; tail -n 3 $O/pytest_fused.txt
python tools/r05/narrow_time.py 2>&1 | tee $O/narrow_time.txt
bash tools/exp/ab_headline.sh "DC_NARROW_CHAIN=0" "DC_NARROW_CHAIN=1" 200 > $O/ab_narrow.txt 2>&1
cat $O/ab_narrow.txt
