#!/bin/bash
# round 6: the driver's commands - GPU suite with -x, smoke, default bench - on one box; outputs under gpurun_out/r06_$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_${1:-check}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -n 12
cp gpurun_out/parity_distances.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.txt
S=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"; tail -c 300 $O/bench.err
cp gpurun_out/bench_kernel_stats.csv $O/ 2>/dev/null
python - "$O" <<'PY'
import json, sys
j=json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], j["ms_per_step_windows"]["min"], j["ms_per_step_windows"]["median"])
d=j.get("dropin_reference_wiring") or {}
print("dropin", d.get("ms_per_step"), d.get("ratio_to_headline"), "one-stream enc", (d.get("contact_encoder_one_stream") or {}).get("ms_per_step"), "branch streams", json.dumps(d.get("branch_streams_opt_in"))[:160])
r=j["roofline"]; print("roofline frac", r["frac"], "achieved", r["achieved"], "ko", r["kernel_only_avg_launch_us"], "live", r["frac_live_hip_events"], "work-eq", r["frac_work_equivalent_per_hop_compulsory"], "traffic", r["traffic"], r.get("counter_over_min_traffic_ratio"))
m=j["roofline_mfma"]; print("mfma frac", m["frac"], m.get("frac_kernel_only"), m.get("kernel_only_us_per_6_launches"))
for k in ("strict_fp32","merged_branches","full_train_step_b4","full_train_step_b32","train_loop_b4"):
    v=j.get(k) or {}; print(k, v.get("value"), v.get("ms_per_step"))
PY
