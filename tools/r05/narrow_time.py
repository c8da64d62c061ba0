#!/usr/bin/env python3
"""Device time of a first layer's input slab (pack + 3 narrow hops) for the B=32 everyday batch: one dc_hop_chain_narrow_f32
launch against dc_spmm_f32_pack + 2 x dc_spmm_f32, replayed from a hipGraph (20 calls per replay), HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402

dev = torch.device("cuda:0")
rest, _, rig = (b.to(dev) for b in synth.make_batch(32))
for name, b in (("soft F=21", rest), ("rigid F=25", rig)):
    g = GraphIndex(b.edge_index, b.x.size(0), segments=b.segments())
    x = b.x
    for narrow in (True, False):
        ops.NARROW_CHAIN = narrow
        slab, _ = ops._build_input_slab(g, x, 3, False)
        into = (slab, None)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                ops._build_input_slab(g, x, 3, False, into=into)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    ops._build_input_slab(g, x, 3, False, into=into)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            torch.cuda.synchronize()
            e0.record(s)
            for _ in range(10):
                gr.replay()
            e1.record(s)
            torch.cuda.synchronize()
        print(f"{name}: {'one launch (narrow chain)' if narrow else 'pack + 2 hops (3 launches)'}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per slab")
