#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_reduce; mkdir -p $O
{
python tools/r06/dw_narrow_time.py 2>&1 | grep -v amdgpu.ids
python tools/r06/dense_time.py 2>&1 | grep -v amdgpu.ids | grep dW
timeout 900 python -m pytest tests/test_wide_dense.py tests/test_gpu_parity.py tests/test_merged.py tests/test_bf16_backward.py tests/test_attention_flash.py tests/test_dw_order.py -x -q -m gpu 2>&1 | tail -n 3
timeout 200 python tools/r06/skip_probe.py base 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  two streams: \1 ms/'
timeout 200 python tools/r06/skip_probe.py base --serial-branches 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  one stream: \1 ms/'
} | tee $O/reduce.txt
