#!/bin/bash
# kernel statistics of the whole reference train step at the SHIPPED batch 4 (configs/everyday.json:26), eager
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_b4
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/full_step.py --batch 4 --steps 30 --graph > $O/log.txt 2>&1
tail -n 2 $O/log.txt
f=$(ls $O/stats/*/*_kernel_stats.csv | head -1)
cp $f $O/kernel_stats.csv
rm -rf $O/stats
head -n 40 $O/kernel_stats.csv | cut -c1-140
