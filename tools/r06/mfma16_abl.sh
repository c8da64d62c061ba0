#!/bin/bash
# round 6: timing-only ablation - k_fwd_h2d with two 16x16x32 MFMAs in place of every 32x32x16 (same FLOPs / reads / VALU)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_mfma16
mkdir -p $O
for i in 1 2; do
  echo "== product library"; python tools/r06/dense_time.py 2>&1 | grep -v amdgpu.ids
  echo "== variant: 16x16x32 (timing only)"; python tools/exp/run_with_lib.py tools/r06/lib_mfma16.so tools/r06/dense_time.py 2>&1 | grep -v amdgpu.ids
done | tee $O/mfma16_abl.txt
