#!/bin/bash
# round 6: drop-in tests, the GPU suite, the default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_a
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dropin.py -x -q -m gpu > $O/pytest_dropin.txt 2>&1; echo "dropin rc=$?"; tail -n 15 $O/pytest_dropin.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -n 8 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r06_a/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], j["ms_per_step_windows"])
print("dropin", json.dumps(j.get("dropin_reference_wiring"))[:600])
r=j["roofline"]; print("roofline frac", r["frac"], "achieved", r["achieved"], "ko", r["kernel_only_avg_launch_us"], "live", r["frac_live_hip_events"], "work-eq", r["frac_work_equivalent_per_hop_compulsory"], r["per_kernel"])
m=j["roofline_mfma"]; print("mfma frac", m["frac"], m.get("frac_kernel_only"), m.get("kernel_only_us_per_6_launches"))
PY
cp gpurun_out/bench_kernel_stats.csv $O/ 2>/dev/null
