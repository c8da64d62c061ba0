#!/usr/bin/env python3
"""Does the F = 256 hop run faster as 2 (4) launches over column halves (quarters)?  An XCD's share of a batch-32 soft graph
is 4 meshes = 4.2 MB of feature rows - the size of its L2; a column half is 2.1 MB.  Chains of 3 hops over a step-shaped
slab (x_k -> x_{k+1}), graph-replayed, us per hop of the chain."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rest, _, rig = synth.make_batch(32)
    for name, b in (("soft", rest), ("rigid", rig)):
        g = GraphIndex(b.edge_index.to(dev), b.x.shape[0])
        n = b.x.shape[0]
        slabs = [ops._alloc_slab(n, 1024, dev) for _ in range(6)]       # rotate: no slab stays cache-resident by accident
        for s in slabs:
            s.normal_()
        for split in (1, 2, 4):
            w = 256 // split

            def chain(s):
                for k in range(3):
                    for c in range(split):
                        ops.hop(g.fwd, s[:, k * 256 + c * w:k * 256 + (c + 1) * w],
                                out=s[:, (k + 1) * 256 + c * w:(k + 1) * 256 + (c + 1) * w])
            chain(slabs[0])
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for s in slabs:
                    chain(s)
            gr.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (10 * len(slabs) * 3) * 1e3
            print(f"{name:5s} N={n} F=256 hop as {split} launch(es) of {w:3d} columns: {us:6.2f} us per hop")


if __name__ == "__main__":
    main()
