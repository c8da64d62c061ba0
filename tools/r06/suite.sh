#!/bin/bash
# round 6: the whole GPU suite WITHOUT -x (every failure at once), then with the driver's flags if green
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_${1:-suite}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -n 30
cp gpurun_out/parity_distances.json $O/ 2>/dev/null
