#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_merged
mkdir -p $O
timeout 600 python -m pytest tests/test_merged.py tests/test_hop_chain.py -x -q -m gpu 2>&1 | tail -n 5
timeout 300 python bench.py --no-pmc --no-kernel-trace --no-strict-fp32 --no-backbones --no-radius100k --no-full-step --no-cpu-baseline --no-dropin --windows 0 --settle 50 > $O/bench_merged.json 2>$O/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r06_merged/bench_merged.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"]); print("merged", json.dumps(j.get("merged_branches"))[:700])
PY
