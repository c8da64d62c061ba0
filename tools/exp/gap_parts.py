#!/usr/bin/env python3
"""What the "new batch every step" gap of bench.py is made of: the encoder step (fwd + bwd + Adam, one hipGraph) at B = 32
  A  one fixed batch, topology built once outside the graph                (bench: value_cached_topology)
  B  A + the eager packed copy of a new batch in front of every replay     (copy + eager-kernel -> graph transition)
  C  copy eager, topology (adjacency build + first-layer hops) in the graph (bench: value)
  D  C with the copy captured as the graph's first node (one graph per staging buffer)
ms per step each, 200 replays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd import graph as dc_graph  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    nb, B = 4, 32
    pool, layout, static_buf = [], None, None
    for j in range(nb):
        r_h, _, g_h = synth.make_batch(B, first_idx=j * B)
        tensors = (r_h.x, r_h.edge_index, g_h.x, g_h.edge_index)
        if layout is None:
            layout, off = [], 0
            for t in tensors:
                layout.append((off, t.numel() * t.element_size(), t.dtype, tuple(t.shape)))
                off += (t.numel() * t.element_size() + 255) // 256 * 256
            static_buf = torch.empty(off, dtype=torch.uint8, device=dev)
        packed = torch.empty_like(static_buf)
        for t, (off, nbytes, dt, shp) in zip(tensors, layout):
            packed[off:off + nbytes].view(dt).view(shp).copy_(t.to(dev))
        pool.append(packed)
        if j == 0:
            rest, rig = r_h.to(dev), g_h.to(dev)
            lay = rest.segments(), rig.segments()
            rest.x, rest.edge_index, rig.x, rig.edge_index = [static_buf[o:o + n].view(dt).view(shp) for o, n, dt, shp in layout]
            rest.assume_segments(lay[0])
            rig.assume_segments(lay[1])
            static_buf.copy_(packed)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    g_rest = torch.randn(rest.x.shape[0], 256, device=dev)
    g_rig = torch.randn(rig.x.shape[0], 256, device=dev)
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    bucket.zero()

    def body():
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [g_rest, g_rig])
        bucket.all_reduce_mean()
        opt.step()

    def warm():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def timeit(step, n=200):
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # A / B: static topology
    dc_graph.clear_cache()
    for g_ in enc.topology(rest, rig):
        g_._static_ok = True
    warm()
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        body()
    print(f"A cached topology, replay only             : {timeit(lambda i: ga.replay()):.4f} ms")
    print(f"B A + eager packed copy before every replay : {timeit(lambda i: (static_buf.copy_(pool[i % nb]), ga.replay())):.4f} ms")
    # C: topology inside
    dc_graph.clear_cache()
    warm()
    dc_graph.clear_cache()
    gc_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc_):
        body()
    print(f"C eager copy + graph with topology work     : {timeit(lambda i: (static_buf.copy_(pool[i % nb]), gc_.replay())):.4f} ms")
    print(f"  (graph with topology work, no copy)       : {timeit(lambda i: gc_.replay()):.4f} ms")
    # D: the copy inside, one graph per staging buffer, shared memory pool
    graphs = []
    for j in range(nb):
        dc_graph.clear_cache()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None):
            static_buf.copy_(pool[j])
            body()
        graphs.append(g)
    print(f"D copy captured in the graph (1 graph/buffer): {timeit(lambda i: graphs[i % nb].replay()):.4f} ms")


if __name__ == "__main__":
    main()
