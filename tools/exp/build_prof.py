#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats -- python3 tools/exp/build_prof.py : per-kernel durations of
dc_graph_build on the two B=32 everyday graphs (50 rebuilds each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402

dev = torch.device("cuda:0")
rest, _, rig = synth.make_batch(32)
for b in (rest, rig):
    g = GraphIndex(b.edge_index.to(dev), b.x.shape[0])
    for _ in range(50):
        g.rebuild()
torch.cuda.synchronize()
