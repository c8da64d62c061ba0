#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter): mean per dispatch.  usage: pmc_sum.py DIR [substr]"""
import csv
import glob
import sys
from collections import defaultdict

acc, cnt = defaultdict(float), defaultdict(int)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub not in k:
            continue
        key = (k[:60], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        cnt[key] += 1
disp = defaultdict(set)
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:32s} sum {v:16.0f} rows {cnt[(k, c)]}")
