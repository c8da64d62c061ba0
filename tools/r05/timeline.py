#!/usr/bin/env python3
"""One graph-replayed bench step as a timeline.  Input: a rocprofv3 --kernel-trace CSV of bench.py; steps are delimited by
consecutive dc::k_adam_flat dispatches.  Prints, for a median-length step of the timed region: every kernel with its start
offset, duration, queue, and the idle time on its queue before it; then per queue the summed kernel time and idle time, and the
time during which NO kernel ran.   python tools/r05/timeline.py <kernel_trace.csv> [longest]"""
import csv
import re
import sys
from collections import Counter


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("dc::", "")
    return name[:44]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "k_adam_flat" in r[2]]
    steps = []
    for a, b in zip(adam[:-1], adam[1:]):
        ks = rows[a + 1:b + 1]
        if ks:
            steps.append((rows[b][1] - rows[a][1], a, b))
    counts = Counter(b - a for _, a, b in steps)
    nk = counts.most_common(1)[0][0]
    if len(sys.argv) > 2 and sys.argv[2] == "longest":       # the step shape with the most kernels among the frequent ones
        nk = max(k for k, c in counts.items() if c >= 20)     # (the headline step: topology work inside)
    sel = sorted(s for s in steps if s[2] - s[1] == nk)
    wall, a, b = sel[len(sel) // 2]
    t0 = rows[a][1]
    ks = rows[a + 1:b + 1]
    print(f"step of {nk} kernels, wall {wall / 1e3:.1f} us (median of {len(sel)} such steps)")
    last_end = {}
    per_q = {}
    for s, e, n, q in ks:
        idle = (s - last_end[q]) / 1e3 if q in last_end else (s - t0) / 1e3
        last_end[q] = e
        d = per_q.setdefault(q, [0.0, 0.0, 0])
        d[0] += (e - s) / 1e3
        d[1] += max(idle, 0.0)
        d[2] += 1
        print(f"  q{q:>3} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  (queue idle before: {idle:6.1f})  {short(n)}")
    busy, cs, ce = 0, None, None
    for s, e, _, _ in ks:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    for q, (kt, idle, n) in per_q.items():
        print(f"queue {q}: {n} kernels, kernel time {kt:.1f} us, idle between its kernels {idle:.1f} us")
    print(f"no kernel running: {(wall - busy) / 1e3:.1f} us of {wall / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
