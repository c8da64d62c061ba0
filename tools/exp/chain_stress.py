#!/usr/bin/env python3
"""Direct stress of the 3-hop chain launch (LDS-table form forced) on small graphs: the same launch on the same inputs over
and over on stream A while stream B runs other library work; every result compared bit for bit with the first one.
    python tools/exp/chain_stress.py [iterations] [other: rigid | dense | attn | none]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    other = sys.argv[2] if len(sys.argv) > 2 else "rigid"
    dev = torch.device("cuda:0")
    ops.HOP_CHAIN_GCN_MIN_NODES = int(os.environ.get("STRESS_MIN_NODES", "0"))
    rest, _, rig = (b.to(dev) for b in synth.make_batch(2, soft_vertices=256, sphere_resolution=8))
    g = GraphIndex(rest.edge_index, rest.x.shape[0], segments=rest.segments())
    n, f, k = rest.x.shape[0], 256, 3
    torch.manual_seed(1)
    slab = ops._alloc_slab(n, (k + 1) * f, dev)
    slab[:, :f] = torch.randn(n, f, device=dev) * 1e-3
    rm = torch.zeros(n, device=dev)

    def chain():
        rm.zero_()
        ops.chained_hops(g, slab, f, k, backward=False, rowmax=rm, transposed=True, rowmax_zeroed=True)
    chain()
    torch.cuda.synchronize()
    want, want_rm = slab.clone(), rm.clone()
    # the co-runner on stream B
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    gb = torch.randn(rig.x.shape[0], 256, device=dev)
    a_ = torch.randn(512, 256, device=dev)
    b_ = torch.randn(381, 256, device=dev)

    def co():
        if other == "rigid":                       # the rigid encoder branch, forward + backward (tiny dense / dW / generic kernels)
            out = enc._branch(enc.conv_layers_rigid, rig.x, rig.edge_index, rig.segments())
            out.backward(gb)
        elif other == "dense":
            (ops.dense_linear(a_, torch.randn(256, 256, device=dev), None, relu=True)).sum()
        elif other == "attn":
            (torch.softmax(a_ @ b_.t(), dim=-1) @ b_).sum()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    keep, bad = [], 0
    for i in range(iters):
        with torch.cuda.stream(sb):
            if other != "none":
                co()
        with torch.cuda.stream(sa):
            chain()
            keep.append((slab.clone(), rm.clone()))
        if len(keep) == 64 or i + 1 == iters:
            torch.cuda.synchronize()
            for s_, r_ in keep:
                if not torch.equal(s_, want) or not torch.equal(r_, want_rm):
                    bad += 1
                    d = (s_ != want)
                    rows = d.any(1).nonzero().flatten()[:8].tolist()
                    cols = d.any(0).nonzero().flatten()[:8].tolist()
                    print(f"  mismatch: {int(d.sum())} elements, rows {rows}, columns {cols}, row maxima equal {torch.equal(r_, want_rm)}", flush=True)
            keep = []
    print(f"co-runner {other}: {bad} of {iters} chain launches differ from the first one")


if __name__ == "__main__":
    main()
