#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_turnstile; mkdir -p $O
for rep in 1 2; do
for t in 0 1 2; do
  DC_TURNSTILE=$t timeout 200 python tools/r06/skip_probe.py base 2>&1 | grep -E "ms_per_step" | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*/DC_TURNSTILE=$t two streams: \1 ms/"
done
done | tee $O/turnstile_ab.txt
DC_TURNSTILE=1 timeout 300 python -m pytest tests/test_dropin.py tests/test_merged.py tests/test_full_size.py -x -q -m gpu 2>&1 | tail -n 4
