#!/usr/bin/env python3
"""Can the per-batch topology work (adjacency build + first-layer hop slabs of the NEXT batch) hide under the current step?
B = 32 everyday batch, encoder fwd + bwd + Adam as one hipGraph over double-buffered static inputs, three forms:
  cached      adjacency and first-layer slabs constant (built once, outside)                                  [lower bound]
  serial      every replay first rebuilds its own slot's adjacency + slabs (what bench.py's headline step does)
  overlapped  every replay rebuilds the OTHER slot's adjacency + slabs on a third stream forked where the backward starts (what
              a prefetching loader does; round 2 measured no gain from a third branch forked at the step's START)
The slabs are the eagerly cached ones, recomputed in place (`ops._build_input_slab(into=)`): the cache entries are never
re-filed, so no captured address can dangle (cf. profiles/r05/j_refresh_mode_memory_fault.txt).
python tools/r05/prep_overlap.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, ops, synth  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))


def slot(first):
    r, _, g = synth.make_batch(B, first_idx=first)
    return r.to(dev), g.to(dev)


slots = [slot(0), slot(B)]
torch.manual_seed(0)
enc = ContactEncoder([21, 25], 256).to(dev)
g_rest = torch.randn(slots[0][0].x.shape[0], 256, device=dev)
g_rig = torch.randn(slots[0][1].x.shape[0], 256, device=dev)
bucket = dp.GradBucket(enc.parameters(), direct=True)
opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
bucket.zero()
prep_stream = torch.cuda.Stream()

# eager: both slots' adjacency and first-layer slabs, constant objects for the life of the graphs
topos = [enc.topology(*slots[s]) for s in (0, 1)]
slabs = []
for s in (0, 1):
    per = []
    for g, b in zip(topos[s], slots[s]):
        ops.precompute_input_hops(g, b.x, 3)
        g._static_ok = True
        (entry,) = g._hop_cache.values()
        per.append((entry[0], entry[1]))
    slabs.append(per)
torch.cuda.synchronize()


def prep(s):
    """slot s: adjacency rebuilt in place, first-layer slabs recomputed in place"""
    for g, b, into in zip(topos[s], slots[s], slabs[s]):
        g.rebuild()
        ops._build_input_slab(g, b.x, 3, False, into=into)


def train(s, fork=None):
    a, b = enc(*slots[s])
    if fork is not None:
        fork()
    torch.autograd.backward([a, b], [g_rest, g_rig])
    bucket.all_reduce_mean()
    opt.step()


def capture(body):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    torch.cuda.synchronize()
    return g


def timed(graphs, reps=300):
    for i in range(6):
        graphs[i & 1].replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        graphs[i & 1].replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def body_serial(s):
    prep(s)
    train(s)


def body_overlapped(s):
    main = torch.cuda.current_stream()

    def fork():
        prep_stream.wait_stream(main)
        with torch.cuda.stream(prep_stream):
            prep(1 - s)
    train(s, fork)
    main.wait_stream(prep_stream)


keep = []
for name, body in (("cached", lambda s: train(s)), ("serial", body_serial), ("overlapped", body_overlapped)):
    graphs = [capture(lambda s=s: body(s)) for s in (0, 1)]
    keep.append(graphs)                                # (the graphs stay alive: nothing they reference is freed)
    print(f"{name:11s}: {timed(graphs) * 1e3:7.1f} us per step", flush=True)
