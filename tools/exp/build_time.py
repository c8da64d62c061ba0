#!/usr/bin/env python3
"""dc_graph_build (both sides + gcn_norm) on the B=32 graphs (or `build_time.py B`), graph-replayed: us per build (soft,
rigid, merged; and the one-launch segmented build of the same batches).  DC_CSR_BUCKETS=0 / 1 forces the windowed /
the bucketed pipeline for the three global builds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rest, _, rig = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    parts = [(rest.edge_index.to(dev), rest.x.shape[0]), (rig.edge_index.to(dev), rig.x.shape[0])]
    seg = [GraphIndex(*parts[0], segments=rest.segments()), GraphIndex(*parts[1], segments=rig.segments())]
    assert all(g._segments is not None for g in seg)
    for name, g in (("soft", GraphIndex(*parts[0])), ("rigid", GraphIndex(*parts[1])),
                    ("merged", GraphIndex.from_parts(parts)), ("soft/1 launch", seg[0]), ("rigid/1 launch", seg[1])):
        for _ in range(3):
            g.rebuild()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                g.rebuild()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:14s} N={g.num_nodes:6d} E={g.num_input_edges:7d}: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us per build")


if __name__ == "__main__":
    main()
