#!/usr/bin/env python3
"""Round 5: where does the rare run-to-run difference of the LDS-table chain launch (profiles/r04/e_chain_rerun_difference.txt)
come from - the kernel, or what it reads?  Same loop as dp_flake2.py (eager train step, new batch tensors every step, two
streams), with the LDS-table form forced on 256-vertex meshes, plus:

  HUNT_DOUBLE=1   right after every chain launch, run the SAME launch again from a copy of the source block into a second slab
                  (same stream) and count differing elements on the device: kernel non-determinism shows here
  HUNT_TAPS=1     clone the gradient slab right after the chain launch and again after the dX block; report which differ
                  between repetitions, element by element (block, row, column, values), with the mesh neighbourhood
python tools/exp/chain_hunt.py [repeats]"""
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
DOUBLE = os.environ.get("HUNT_DOUBLE") == "1"
TAPS = os.environ.get("HUNT_TAPS") == "1"
if os.environ.get("HUNT_SERIAL") == "1":
    ContactEncoder.overlap_branches = False

rec = {}
dbl = {"count": None, "launches": 0, "keep": []}


def chain_hook(g, adj, slab, f, k, transposed):
    name = f"{'T' if transposed else 'F'}{slab.size(0)}"
    if DOUBLE:
        base = slab._base if slab._base is not None else slab
        again = torch.empty_like(base)
        v = again[:, :slab.size(1)]
        v[:, :f].copy_(slab[:, :f])
        ops.hop_chain(g, adj, v, f, k, weighted=g.normalize)
        ne = (v[:, :(k + 1) * f] != slab[:, :(k + 1) * f]).sum()
        dbl["count"] = ne if dbl["count"] is None else dbl["count"] + ne
        dbl["launches"] += 1
    if TAPS:
        kname, i = "chain." + name, 0
        while kname in rec:
            i += 1
            kname = f"chain.{name}#{i}"
        rec[kname] = (slab[:, :(k + 1) * f].detach().clone(), adj, g)


def tap(name, t):
    if not TAPS or not name.endswith("gslab"):
        return
    kname, i = name, 0
    while kname in rec:
        i += 1
        kname = f"{name}#{i}"
    rec[kname] = (t.detach().clone(), None, None)


def batch(step, dev):
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=step * B, soft_vertices=SV, sphere_resolution=SR))


def run(init, dev):
    m = load_model(EVERYDAY_NETWORK).to(dev)
    m.load_state_dict(init)
    bk = dp.GradBucket(m.parameters(), direct=True)
    op = dp.FlatAdam(bk, lr=4e-4, zero_grad_in_step=True)
    bk.zero()
    snaps = []
    for s in range(STEPS):
        rec.clear()
        losses(m, *batch(s, dev), 1.0)["loss"].backward()
        bk.wait_direct_writes()
        grads = bk.flat.clone()
        op.step()
        torch.cuda.synchronize()
        snaps.append((grads, {k: (v[0], v[1]) for k, v in rec.items()}))
    return snaps


def report(k, a, b, adj):
    ne = a != b
    idx = ne.nonzero()
    rows = sorted(set(idx[:, 0].tolist()))
    print(f"    {k}: {int(ne.sum())} elements, {len(rows)} rows, |diff| {float((a - b).abs()[ne].min()):.2e}..{float((a - b).abs().max()):.2e}")
    f = 256
    per_block = {}
    for r, c in idx.tolist():
        per_block.setdefault(c // f, []).append((r, c % f))
    for blk in sorted(per_block):
        el = per_block[blk]
        rs = sorted({r for r, _ in el})
        cs = sorted({c for _, c in el})
        print(f"      block {blk}: rows {rs[:24]}{'...' if len(rs) > 24 else ''} cols {cs[:16]}")
        for r, c in el[:6]:
            print(f"        [{r},{c}] now {float(a[r, blk * f + c]):+.9e} base {float(b[r, blk * f + c]):+.9e} "
                  f"row max {float(b[r, blk * f:(blk + 1) * f].abs().max()):.3e}")
    if adj is not None and len(per_block) > 1:
        ptr, other = adj.ptr.cpu(), adj.other.cpu()
        blks = sorted(per_block)
        for b0, b1 in zip(blks[:-1], blks[1:]):
            r0 = {r for r, _ in per_block[b0]}
            r1 = {r for r, _ in per_block[b1]}
            nb = set()
            for r in r0:
                nb.update(other[int(ptr[r]):int(ptr[r + 1])].tolist())
            # rows of block b1 = A^T rows: row i of block b1 sums block b0 rows other[ptr[i]:ptr[i+1]]
            dep = {i for i in r1 if set(other[int(ptr[i]):int(ptr[i + 1])].tolist()) & r0}
            print(f"      rows differing in block {b1}: {len(r1)}; of them reading a differing row of block {b0}: {len(dep)}")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    ops.DEBUG_CHAIN = chain_hook
    ops.DEBUG_TAP = tap
    os.environ["HUNT_TAP_BIG"] = "1"
    torch.manual_seed(100)
    init = {k: v.detach().clone() for k, v in load_model(EVERYDAY_NETWORK).to(dev).state_dict().items()}
    base = run(init, dev)
    bad = 0
    for rep in range(reps):
        cur = run(init, dev)
        for s in range(STEPS):
            if not torch.equal(cur[s][0], base[s][0]):
                bad += 1
                print(f"rep {rep} step {s}: gradient bucket differs", flush=True)
                for k in cur[s][1]:
                    a, b = cur[s][1][k][0], base[s][1][k][0]
                    if not torch.equal(a, b):
                        report(k, a, b, cur[s][1][k][1])
                break
    print(f"{bad} of {reps} repetitions differ from the first", flush=True)
    if DOUBLE:
        print(f"HUNT_DOUBLE: {int(dbl['count'])} differing elements between back-to-back launches over {dbl['launches']} launches",
              flush=True)


if __name__ == "__main__":
    main()
