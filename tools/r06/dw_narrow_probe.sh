#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_dw; mkdir -p $O
for cfg in "512 128" "1024 256" "2048 512" "512 64"; do
  set -- $cfg
  echo "== DC_DW_TARGET=$1 DC_DW_MAXCHUNKS=$2"
  DC_DW_TARGET=$1 DC_DW_MAXCHUNKS=$2 python tools/r06/dw_narrow_time.py 2>&1 | grep -v amdgpu.ids
done | tee $O/dw_narrow_probe.txt
