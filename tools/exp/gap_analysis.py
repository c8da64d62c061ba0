#!/usr/bin/env python3
"""How busy is the device inside a graph-replayed step?  Input: a rocprofv3 --kernel-trace CSV of bench.py.
Steps are delimited by consecutive dc::k_adam_flat dispatches; per step: wall (end of Adam to end of the next Adam),
time with at least one kernel running, summed kernel time, number of kernels.

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --no-cpu-baseline --no-full-step \
        --no-strict-fp32 --no-radius100k --no-pmc [--serial-branches]
    python tools/exp/gap_analysis.py out/**/*_kernel_trace.csv"""
import csv
import sys


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "k_adam_flat" in r[2]]
    steps = []
    for a, b in zip(adam[:-1], adam[1:]):
        ks = rows[a + 1:b + 1]
        if not ks:
            continue
        t0, t1 = rows[a][1], rows[b][1]
        busy, cur_s, cur_e = 0, None, None
        for s, e, _ in ks:
            s = max(s, t0)
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        steps.append((t1 - t0, busy, sum(e - s for s, e, _ in ks), len(ks)))
    # the timed region: the longest run of steps with the same kernel count
    from collections import Counter
    nk = Counter(s[3] for s in steps).most_common(1)[0][0]
    sel = [s for s in steps if s[3] == nk]
    sel = sel[len(sel) // 4:]                       # drop the warm-up quarter
    n = len(sel)
    wall = sum(s[0] for s in sel) / n / 1e3
    busy = sum(s[1] for s in sel) / n / 1e3
    ksum = sum(s[2] for s in sel) / n / 1e3
    print(f"{path}: {n} steps of {nk} kernels: wall {wall:.1f} us, >=1 kernel running {busy:.1f} us "
          f"({busy / wall:.1%}), idle {wall - busy:.1f} us, summed kernel time {ksum:.1f} us "
          f"(overlap factor {ksum / busy:.2f})")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
