#!/usr/bin/env python3
"""Round 5, third pass (chain_hunt2.py established: correct inputs in memory, wrong blocks from the product's transposed
chain launch of the RIGID branch, a relaunch 50 us later is right).  Which launch fails?  Around every transposed chain launch
of the step: A = the same launch BEFORE it (scratch slab, scratch row maxima, same mode), B = the product's launch, C = after
it with scratch row maxima, D = after it without row maxima.  After the step's synchronisation each is compared with the
hop-by-hop truth (dc_spmm_f32 over the same arrays).  Prints failure counts per position and problem size.
HUNT_SERIAL=1: both encoder branches on one stream.   python tools/exp/chain_hunt3.py [repeats]"""
import collections
import os
import sys

os.environ.setdefault("DC_HOP_CHAIN_GCN_MIN_NODES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, ops, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, ContactEncoder, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402

STEPS, B = 4, int(os.environ.get("HUNT_B", "2"))
SV = int(os.environ.get("HUNT_SV", "256"))
SR = int(os.environ.get("HUNT_SR", "8"))
if os.environ.get("HUNT_SERIAL") == "1":
    ContactEncoder.overlap_branches = False
rec = []
fails = collections.Counter()
launches = collections.Counter()
_rowmax_of = {}


def _scratch(slab, f):
    base = slab._base if slab._base is not None else slab
    again = torch.empty_like(base)
    v = again[:, :slab.size(1)]
    v[:, :f].copy_(slab[:, :f])
    return v


_orig_hop_chain = ops.hop_chain


def hop_chain_spy(g, adj, slab, f, k, weighted=True, rowmax=None, rowmax_mode=0, src_block=0, direction=1):
    _rowmax_of[slab.data_ptr()] = (rowmax, rowmax_mode)
    return _orig_hop_chain(g, adj, slab, f, k, weighted=weighted, rowmax=rowmax, rowmax_mode=rowmax_mode,
                           src_block=src_block, direction=direction)


def pre(g, adj, slab, f, k, transposed):
    if not transposed:
        return
    d = dict(g=g, adj=adj, f=f, k=k, n=slab.size(0), b0=slab[:, :f].clone())
    # A: the launch the product is about to make, on scratch memory (row maxima: a scratch copy of the block-0 maxima
    # is not available here - mode 1 clears its buffer itself; the arithmetic of the blocks is the same)
    v = _scratch(slab, f)
    rm = torch.empty(slab.size(0), dtype=torch.float32, device=slab.device)
    _orig_hop_chain(g, adj, v, f, k, weighted=g.normalize, rowmax=rm, rowmax_mode=1)
    d["A"] = v[:, :(k + 1) * f]
    rec.append(d)


def post(g, adj, slab, f, k, transposed):
    if not transposed:
        return
    d = rec[-1]
    d["B"] = slab[:, :(k + 1) * f].clone()
    v = _scratch(slab, f)
    rm = torch.empty(slab.size(0), dtype=torch.float32, device=slab.device)
    _orig_hop_chain(g, adj, v, f, k, weighted=g.normalize, rowmax=rm, rowmax_mode=1)
    d["C"] = v[:, :(k + 1) * f]
    v2 = _scratch(slab, f)
    _orig_hop_chain(g, adj, v2, f, k, weighted=g.normalize)
    d["D"] = v2[:, :(k + 1) * f]


def truth(d):
    adj, f, k, n = d["adj"], d["f"], d["k"], d["n"]
    s4 = ops._alloc_slab(n, (k + 1) * f, d["b0"].device)
    s4[:, :f].copy_(d["b0"])
    for j in range(k):
        ops.hop(adj, s4[:, j * f:(j + 1) * f], out=s4[:, (j + 1) * f:(j + 2) * f], weighted=True)
    return s4[:, :(k + 1) * f]


def batch(step, dev):
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=step * B, soft_vertices=SV, sphere_resolution=SR))


def run(init, dev, rep):
    m = load_model(EVERYDAY_NETWORK).to(dev)
    m.load_state_dict(init)
    bk = dp.GradBucket(m.parameters(), direct=True)
    op = dp.FlatAdam(bk, lr=4e-4, zero_grad_in_step=True)
    bk.zero()
    for s in range(STEPS):
        rec.clear()
        losses(m, *batch(s, dev), 1.0)["loss"].backward()
        bk.wait_direct_writes()
        op.step()
        torch.cuda.synchronize()
        for d in rec:
            t = truth(d)
            for pos in "ABCD":
                launches[(d["n"], pos)] += 1
                if not torch.equal(d[pos], t):
                    fails[(d["n"], pos)] += 1
                    ne = (d[pos] != t).nonzero()
                    blocks = sorted({int(c) // d["f"] for c in ne[:, 1].tolist()})
                    rows = sorted(set(ne[ne[:, 1] // d["f"] == blocks[0]][:, 0].tolist()))
                    cols = sorted(set((ne[ne[:, 1] // d["f"] == blocks[0]][:, 1] % d["f"]).tolist()))
                    print(f"rep {rep} step {s}: N={d['n']} launch {pos} differs from the hop-by-hop truth: blocks {blocks}, in block "
                          f"{blocks[0]} rows {rows[:10]} cols {cols[:10]}", flush=True)
        torch.cuda.synchronize()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    ops.DEBUG_CHAIN_PRE, ops.DEBUG_CHAIN = pre, post
    torch.manual_seed(100)
    init = {k: v.detach().clone() for k, v in load_model(EVERYDAY_NETWORK).to(dev).state_dict().items()}
    for rep in range(reps):
        run(init, dev, rep)
    print("launches", dict(launches))
    print("failures", dict(fails), flush=True)


if __name__ == "__main__":
    main()
