#!/usr/bin/env python3
"""What the per-cloud preprocessing of the 100k-point radius graph (BASELINE configs[4]) is made of: each piece replayed from
its own hipGraph, HIP events.   python tools/exp/prep_parts.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, NodeOrder, clear_cache  # noqa: E402


def graphed(fn, r=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(r):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / r * 1e3


def main():
    dev = torch.device("cuda:0")
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    pos, ei = pos.to(dev), ei.to(dev)
    n, f = pos.shape[0], 256
    x = torch.randn(n, f, device=dev).bfloat16()
    o = NodeOrder.morton(pos)
    em = o.relabel(ei)
    print(f"N={n} E={ei.shape[1]}")
    print(f"NodeOrder.morton (codes, sort, inverse): {graphed(lambda: NodeOrder.morton(pos)):7.1f} us")
    print(f"relabel edge_index                     : {graphed(lambda: o.relabel(ei)):7.1f} us")

    def build():
        clear_cache()
        return GraphIndex(em, n)
    print(f"GraphIndex (both adjacencies + gcn_norm): {graphed(build):7.1f} us")
    print(f"apply (gather rows, bf16 [N, 256])      : {graphed(lambda: o.apply(x)):7.1f} us")
    print(f"undo                                    : {graphed(lambda: o.undo(x)):7.1f} us")


if __name__ == "__main__":
    main()
