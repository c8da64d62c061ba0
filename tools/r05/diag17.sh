#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag17
mkdir -p $O
for i in 1 2; do
  ( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu_$i.txt 2>&1
  tail -n 3 $O/pytest_gpu_$i.txt
done
( time timeout 700 env HUNT_TAPS=1 python tools/exp/chain_hunt.py 2000 ) > $O/hunt_final_default_streams.txt 2>&1
tail -n 3 $O/hunt_final_default_streams.txt
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
