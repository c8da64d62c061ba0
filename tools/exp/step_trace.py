#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats -- python3 tools/exp/step_trace.py [serial|cached] [--serial-branches]

The bench step run EAGERLY (same kernels as the graph-replayed step, per-kernel durations visible
to the profiler): serial = every step copies a new batch in and rebuilds the adjacency + first-layer
hops; cached = one fixed batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd import graph as dc_graph  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
dev = torch.device("cuda:0")
pool = []
for j in range(2):
    r, _, g = synth.make_batch(32, first_idx=32 * j)
    if j == 0:
        rest, rig = r.to(dev), g.to(dev)
    pool.append((r.x.to(dev), r.edge_index.to(dev), g.x.to(dev), g.edge_index.to(dev)))
torch.manual_seed(0)
enc = ContactEncoder([21, 25], 256).to(dev)
enc.overlap_branches = "--serial-branches" not in sys.argv
g_rest = torch.randn(rest.x.shape[0], 256, device=dev)
g_rig = torch.randn(rig.x.shape[0], 256, device=dev)
bucket = dp.GradBucket(enc.parameters(), direct=True)
opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
bucket.zero()
for i in range(24):
    if mode == "serial":
        xs, es, xr, er = pool[i & 1]
        rest.x.copy_(xs)
        rest.edge_index.copy_(es)
        rig.x.copy_(xr)
        rig.edge_index.copy_(er)
        dc_graph.clear_cache()
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [g_rest, g_rig])
    bucket.all_reduce_mean()
    opt.step()
torch.cuda.synchronize()
