#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_b4joint
mkdir -p $O
for r in 1 2 3; do
  for j in 0 16384; do
    echo -n "round $r DC_ATTN_JOINT_ROWS=$j: "
    DC_ATTN_JOINT_ROWS=$j timeout 300 python tools/full_step.py --batch 4 --steps 100 --graph 2>&1 | tail -n 1
  done
done | tee $O/ab_b4.txt
for j in 0 16384; do echo -n "B=8 DC_ATTN_JOINT_ROWS=$j: "; DC_ATTN_JOINT_ROWS=$j timeout 300 python tools/full_step.py --batch 8 --steps 50 --graph 2>&1 | tail -n 1; done | tee -a $O/ab_b4.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_train_loop.py tests/test_attention_flash.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
