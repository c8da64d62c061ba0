#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag8
mkdir -p $O
( time timeout 900 python -m pytest tests/test_fused_pack.py tests/test_hop_chain.py tests/test_segmented_build.py tests/test_bucketed_build.py tests/test_node_order.py "tests/test_full_size.py::test_default_b32_step_launches_exactly_the_tuned_kernels" -x -q ) > $O/pytest_subset.txt 2>&1
tail -n 6 $O/pytest_subset.txt
bash tools/exp/ab_headline.sh "DC_NARROW_CHAIN=0" "DC_NARROW_CHAIN=1" 200 > $O/ab_narrow.txt 2>&1
cat $O/ab_narrow.txt
for b in 256 384 768; do
  bash tools/exp/ab_headline.sh "DC_DW_BLOCKS=512" "DC_DW_BLOCKS=$b" 200 2>&1 | tee -a $O/ab_dwblocks.txt | tail -n 2
done
