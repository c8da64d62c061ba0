#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_skip
mkdir -p $O
for v in base no_narrow_fwd no_narrow_dw no_narrow_hops no_mask_grad no_weight_prep no_build no_first_layers base; do
  timeout 200 python tools/r06/skip_probe.py $v 2>&1 | grep -E "^==|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  two streams: \1 ms/'
  timeout 200 python tools/r06/skip_probe.py $v --serial-branches 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  one stream:  \1 ms/'
done | tee $O/skip_probe.txt
