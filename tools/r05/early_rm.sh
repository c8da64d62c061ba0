#!/bin/bash
# early_rm.sh: what do the chain kernel's row-maxima atomics cost, and does publishing them before the last hop hide it?
# (variants built by tools/r05/build_variant.sh: norm = -DDC_CHAIN_ABL=256 (no atomics, timing only), early = -DDC_CHAIN_EARLY_RM=1)
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_early_rm; mkdir -p $out
for rep in 1 2; do
  echo "== default build (rep $rep)"; timeout 300 python tools/exp/hop_chain.py 32 2>&1 | grep -v amdgpu.ids
  echo "== early publication (rep $rep)"; timeout 300 python tools/exp/run_with_lib.py tools/r05/lib_early.so tools/exp/hop_chain.py 32 2>&1 | grep -v amdgpu.ids
  echo "== no atomics, timing only (rep $rep)"; SKIP_CHECK=1 timeout 300 python tools/exp/run_with_lib.py tools/r05/lib_norm.so tools/exp/hop_chain.py 32 2>&1 | grep -v amdgpu.ids
done > $out/early_rm.txt 2>&1
echo "== tests of the early build" >> $out/early_rm.txt
DC_SKIP_LAUNCH=1 timeout 600 python tools/exp/run_with_lib.py tools/r05/lib_early.so tools/r05/pytest_main.py tests/test_hop_chain.py -q -m gpu -x 2>&1 | tail -3 >> $out/early_rm.txt
echo "== launch tests (torchrun form included)" >> $out/early_rm.txt
timeout 900 python -m pytest tests/test_z_launch.py -q -m gpu 2>&1 | tail -8 >> $out/early_rm.txt
cat $out/early_rm.txt
