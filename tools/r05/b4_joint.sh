#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_b4joint
mkdir -p $O
for r in 1 2 3; do
  for cfg in "0 0" "16384 0" "16384 1"; do
    set -- $cfg
    echo -n "round $r DC_ATTN_JOINT_ROWS=$1 DC_ATTN_BATCHED_HEADS=$2: "
    DC_ATTN_JOINT_ROWS=$1 DC_ATTN_BATCHED_HEADS=$2 timeout 300 python tools/full_step.py --batch 4 --steps 100 --graph 2>&1 | tail -n 1 | cut -c1-60
  done
done | tee $O/ab_b4_heads.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_train_loop.py tests/test_attention_flash.py tests/test_full_size.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
