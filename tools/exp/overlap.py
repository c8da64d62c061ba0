#!/usr/bin/env python3
"""python tools/exp/overlap.py <kernel_trace.csv>: how much of the topology kernels' time (k_count,
k_fill, k_emit, k_scan*, k_init, k_spmm_sub, k_pack_input) runs while a dense / wide-hop kernel of
another queue is executing."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
prep = [k for k in ks if any(t in k[2] for t in ("k_count", "k_fill", "k_emit", "k_scan", "k_init", "k_spmm_sub", "k_pack_input"))]
big = [k for k in ks if any(t in k[2] for t in ("k_fwd_h2", "k_dw_split", "k_spmm_wave", "k_fwd_split", "k_mask_grad"))]
tot = sum(e - s for s, e, _, _ in prep)
ov = 0
j = 0
for s, e, _, q in prep:
    for bs, be, _, bq in big:
        if be <= s:
            continue
        if bs >= e:
            break
        if bq != q:
            ov += max(0, min(e, be) - max(s, bs))
print(f"prep kernels: {len(prep)}, total {tot / 1e3:.1f} us; overlapped with a big kernel on another queue: {ov / 1e3:.1f} us (sum over pairs)")
print("queues:", sorted({k[3] for k in ks}))
t0 = ks[len(ks) // 2][0]
for s, e, n, q in ks[len(ks) // 2: len(ks) // 2 + 90]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {n[:60]}")
