#!/usr/bin/env python3
"""Round 5: the chain launch's rare difference needs the other branch's kernels on the SAME compute units
(chain_hunt_cumask.py: 35 of 2000 with shared CUs, 0 of 2000 with disjoint CU masks).  Which kernel?  The LDS-table chain
launch of the small rigid batch (2 spheres of 114 nodes, transposed, row maxima) runs on one stream with NEW inputs every
launch (a stale or foreign LDS read cannot pass for the right data) while the other stream loops ONE candidate workload
(OTHER=): attn_stock (softmax(q k^T) v forward + backward on stock PyTorch: rocBLAS GEMMs + ATen softmax, the shapes of the
small test configuration), gemm (rocBLAS q k^T only), softmax (ATen softmax only), dense (this library's dense_linear
forward + backward), none.  Every output is compared with the hop-by-hop result (dc_spmm_f32) of the same input.
python tools/exp/chain_coresident.py [launches]"""
import os
import sys

os.environ.setdefault("DC_HOP_CHAIN_GCN_MIN_NODES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import graph_index  # noqa: E402


def main():
    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    other = os.environ.get("OTHER", "attn_stock")
    dev = torch.device("cuda:0")
    _, _, rig = (x.to(dev) for x in synth.make_batch(2, soft_vertices=256, sphere_resolution=8))
    g = graph_index(rig.edge_index, rig.x.size(0), segments=rig.segments())
    n, f, k = rig.x.size(0), 256, 3
    side, main_s = torch.cuda.Stream(), torch.cuda.Stream()
    ns, nr, d = 512, 228, 256
    q = torch.randn(ns, d, device=dev, requires_grad=True)
    kk = torch.randn(nr, d, device=dev, requires_grad=True)
    v = torch.randn(nr, d, device=dev, requires_grad=True)
    w = torch.randn(d, d, device=dev, requires_grad=True)
    go = torch.randn(ns, d, device=dev)

    def load():
        if other == "attn_stock":
            o = torch.softmax(q @ kk.t(), dim=-1) @ v
            o.backward(go)
            q.grad = kk.grad = v.grad = None
        elif other == "gemm":
            with torch.no_grad():
                for _ in range(4):
                    q @ kk.t()
        elif other == "softmax":
            with torch.no_grad():
                s = q[:, :nr]
                for _ in range(4):
                    torch.softmax(s, dim=-1)
        elif other == "dense":
            y = ops.dense_linear(q, w, None, relu=True)
            y.backward(go)
            q.grad = w.grad = None

    bad = 0
    gen = torch.Generator(device=dev).manual_seed(1)
    for it in range(loops):
        with torch.cuda.stream(side):
            src = torch.randn(n, f, device=dev, generator=gen) * 1e-5
            slab = ops._alloc_slab(n, (k + 1) * f, dev)
            slab[:, :f].copy_(src)
            rm = torch.zeros(n, dtype=torch.float32, device=dev)
        side.wait_stream(torch.cuda.current_stream())
        main_s.wait_stream(side)
        with torch.cuda.stream(main_s):
            for _ in range(3):
                load()
        with torch.cuda.stream(side):
            ops.hop_chain(g, g.bwd, slab, f, k, weighted=True, rowmax=rm, rowmax_mode=2)
        torch.cuda.synchronize()
        ref = ops._alloc_slab(n, (k + 1) * f, dev)
        ref[:, :f].copy_(src)
        for j in range(k):
            ops.hop(g.bwd, ref[:, j * f:(j + 1) * f], out=ref[:, (j + 1) * f:(j + 2) * f], weighted=True)
        if not torch.equal(ref[:, :(k + 1) * f], slab[:, :(k + 1) * f]):
            bad += 1
            ne = (ref[:, :(k + 1) * f] != slab[:, :(k + 1) * f]).nonzero()
            print(f"launch {it}: {ne.size(0)} elements differ, rows {sorted(set(ne[:, 0].tolist()))[:8]} cols "
                  f"{sorted(set((ne[:, 1] % f).tolist()))[:8]}", flush=True)
    print(f"OTHER={other}: {bad} of {loops} chain launches differ from the hop-by-hop result", flush=True)


if __name__ == "__main__":
    main()
