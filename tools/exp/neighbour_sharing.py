#!/usr/bin/env python3
"""How many DISTINCT neighbour rows do runs of consecutive rows of the Morton-ordered 100k-point radius graph (BASELINE
configs[4]) gather?  CPU only (numpy).  Sizing of a hop that stages the union of a run's neighbour rows in LDS once
(DESIGN.md section 8, item 6): fetches per run = distinct rows instead of edges; the largest union must fit the LDS.

    python tools/exp/neighbour_sharing.py [row_bytes=512]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from deformcontact_amd import synth  # noqa: E402


def spread10(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def main():
    row_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    pos, ei = pos.numpy(), ei.numpy()
    lo, hi = pos.min(0), pos.max(0)
    q = np.minimum(((pos - lo) / (hi - lo) * 1024).astype(np.uint64), 1023)       # dc_order.hip: k_codes
    code = spread10(q[:, 0]) | (spread10(q[:, 1]) << 1) | (spread10(q[:, 2]) << 2)
    order = np.argsort(code, kind="stable")
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    src, dst = inv[ei[0]], inv[ei[1]]
    o = np.argsort(dst, kind="stable")
    src, dst = src[o], dst[o]
    n = pos.shape[0]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))])
    deg = np.diff(ptr)
    print(f"N={n} E={len(src)}: in-degree mean {deg.mean():.1f}, max {deg.max()}; {int((deg >= 30).sum())} nodes with >= 30 "
          f"neighbours hold {int(deg[deg >= 30].sum())} edges")
    for run in (8, 32, 64, 128, 256):
        edges = distinct = largest = dense_e = dense_d = 0
        for r0 in range(0, n, run):
            s = src[ptr[r0]:ptr[min(r0 + run, n)]]
            d = len(np.unique(s))
            edges += len(s)
            distinct += d
            largest = max(largest, d)
            if len(s) >= run * 24:
                dense_e += len(s)
                dense_d += d
        print(f"runs of {run:4d} rows: {distinct:7d} row fetches for {edges} edges ({edges / distinct:.2f} x fewer); dense runs "
              f"{dense_e / max(dense_d, 1):.2f} x; largest union {largest} rows = {largest * row_bytes / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
