#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_diag6
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 env HUNT_TAPS=1 python tools/exp/run_with_lib.py tools/r05/lib_poisonend.so tools/exp/chain_hunt.py 3000 ) > $O/hunt_poisonend.txt 2>&1
grep -c "bucket differs" $O/hunt_poisonend.txt; grep -c -i "nan" $O/hunt_poisonend.txt; tail -n 4 $O/hunt_poisonend.txt
# timeline of the graph-replayed step (kernel trace only)
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones --kernel-reps 4 > $O/bench_trace.json 2> $O/bench_trace.err
f=$(ls $O/trace/*/*_kernel_trace.csv 2>/dev/null | head -1)
python tools/r05/timeline.py $f > $O/timeline.txt 2>&1
python tools/exp/gap_analysis.py $f > $O/gaps.txt 2>&1
rm -rf $O/trace
tail -n 12 $O/timeline.txt
