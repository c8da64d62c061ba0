#!/usr/bin/env python3
"""Isolated stress of the launch that fails in the train step: the LDS-table chain kernel on the RIGID batch of the small test
configuration (2 spheres of 114 nodes: k_hop_chain_gcn<1>, 16 workgroups), transposed, with row maxima, launched LOOPS times on a
side stream with fixed inputs while the main stream runs (MAIN=) soft_bwd: the soft branch's backward kernels (mask + chain +
dX block + dW), dense: a wide dense block, chain: the soft chain launch, none.  Every output is compared with the first.
python tools/exp/chain_stress228.py [loops]"""
import os
import sys

os.environ.setdefault("DC_HOP_CHAIN_GCN_MIN_NODES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import graph_index  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402


def main():
    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    mode = os.environ.get("MAIN", "soft_bwd")
    sv, sr, b = int(os.environ.get("HUNT_SV", "256")), int(os.environ.get("HUNT_SR", "8")), int(os.environ.get("HUNT_B", "2"))
    dev = torch.device("cuda:0")
    rest, _, rig = (x.to(dev) for x in synth.make_batch(b, soft_vertices=sv, sphere_resolution=sr))
    g = graph_index(rig.edge_index, rig.x.size(0), segments=rig.segments())
    n, f, k = rig.x.size(0), 256, 3
    torch.manual_seed(0)
    src = torch.randn(n, f, device=dev) * 1e-5
    side = torch.cuda.Stream()
    # main-stream load
    enc = ContactEncoder([21, 25], 256).to(dev)
    gs = graph_index(rest.edge_index, rest.x.size(0), segments=rest.segments())
    ns = rest.x.size(0)
    xs = torch.randn(ns, f, device=dev, requires_grad=True)
    gout = torch.randn(ns, f, device=dev)
    conv = enc.conv_layers_resting[1]

    def main_load():
        if mode == "none":
            return
        if mode == "soft_bwd":
            y = conv(xs, rest.edge_index, relu=True)
            y.backward(gout)
            xs.grad = None
        elif mode == "dense":
            with torch.no_grad():
                ops.dense_linear(torch.randn(32768, 256, device=dev), conv.lins[0].weight, None)
        elif mode == "chain":
            s = ops._alloc_slab(ns, (k + 1) * f, dev)
            ops.hop_chain(gs, gs.bwd, s, f, k)

    def launch():
        slab = ops._alloc_slab(n, (k + 1) * f, dev)
        slab[:, :f].copy_(src)
        rm = torch.zeros(n, dtype=torch.float32, device=dev)
        ops.hop_chain(g, g.bwd, slab, f, k, weighted=True, rowmax=rm, rowmax_mode=2)
        return slab[:, :(k + 1) * f]

    with torch.cuda.stream(side):
        ref = launch().clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(loops):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            outs = [launch() for _ in range(4)]
        main_load()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                ne = (o != ref).nonzero()
                print(f"iteration {it}: differs in {ne.size(0)} elements, rows {sorted(set(ne[:, 0].tolist()))[:10]}", flush=True)
    print(f"MAIN={mode}: {bad} of {loops * 4} launches differ", flush=True)


if __name__ == "__main__":
    main()
