#!/usr/bin/env python3
"""Narrowing the rare mismatch of tests/test_a_dp_graphed.py: the eager train step (losses + backward into a direct-gradient
bucket + FlatAdam) run TWICE from the same state on the same batches, parameters compared after every step.  Environment
switches select what to rule out:  HUNT_SERIAL=1 (no two-stream branches), HUNT_DIRECT=0 (autograd-accumulated gradients),
DC_HOP_CACHE=0, DC_FUSED_PACK=0, DC_FUSED_LOSS=0, DC_HOP_CHAIN=0, DC_HOP_CHAIN_GCN_MIN_NODES=0 (the LDS-table form of
the chain kernel also on small graphs: what shows the difference), HUNT_SYNC=1 (device synchronize before Adam), HUNT_TAP=1
(clones of the backward's row maxima; HUNT_TAP_BIG=1: of the gradient slab and dX too), HUNT_B / HUNT_SV / HUNT_SR (batch and mesh sizes).
python tools/exp/dp_flake2.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, ContactEncoder, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402

STEPS, B = 4, int(os.environ.get("HUNT_B", "2"))
SV = int(os.environ.get("HUNT_SV", "256"))
SR = int(os.environ.get("HUNT_SR", "8"))
if os.environ.get("HUNT_SERIAL") == "1":
    ContactEncoder.overlap_branches = False
DIRECT = os.environ.get("HUNT_DIRECT", "1") != "0"
SYNC = os.environ.get("HUNT_SYNC") == "1"


def batch(step, dev):
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=step * B, soft_vertices=SV, sphere_resolution=SR))


def run(init, dev):
    m = load_model(EVERYDAY_NETWORK).to(dev)
    m.load_state_dict(init)
    bk = dp.GradBucket(m.parameters(), direct=DIRECT)
    op = dp.FlatAdam(bk, lr=4e-4, zero_grad_in_step=True)
    bk.zero()
    snaps = []
    rec = {}
    if os.environ.get("HUNT_HOOKS") == "1":
        for name, layers in (("rest", m.conv_layers_resting), ("rig", m.conv_layers_rigid)):
            for i, l in enumerate(layers):
                l.register_forward_hook(lambda mod, inp, out, k=f"{name}.{i}.out": rec.__setitem__(k, out.detach().clone()))
                l.register_full_backward_hook(
                    lambda mod, gi, go, k=f"{name}.{i}.grad_out": rec.__setitem__(k, go[0].detach().clone()))
    if os.environ.get("HUNT_TAP") == "1":
        from deformcontact_amd import ops

        def tap(name, t):
            k, i = name, 0
            while k in rec:
                i += 1
                k = f"{name}#{i}"
            rec[k] = t.detach().clone()
        ops.DEBUG_TAP = tap
    for s in range(STEPS):
        rec.clear()
        losses(m, *batch(s, dev), 1.0)["loss"].backward()
        if SYNC:
            torch.cuda.synchronize()
        bk.wait_direct_writes()
        grads = bk.flat.clone()
        op.step()
        torch.cuda.synchronize()
        snaps.append((grads, {n: p.detach().clone() for n, p in m.named_parameters()}, dict(rec)))
    return snaps, m, bk


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    torch.manual_seed(100)
    init = {k: v.detach().clone() for k, v in load_model(EVERYDAY_NETWORK).to(dev).state_dict().items()}
    base, m0, bk0 = run(init, dev)
    names = [n for n, _ in m0.named_parameters()]
    bad = 0
    for rep in range(reps):
        cur, m, bk = run(init, dev)
        for s in range(STEPS):
            gd = not torch.equal(cur[s][0], base[s][0])
            pd = [n for n in names if not torch.equal(cur[s][1][n], base[s][1][n])]
            if gd or pd:
                bad += 1
                which = []
                if gd:                                    # which parameters' gradient slices differ
                    off = 0
                    for n, p in m.named_parameters():
                        k = p.numel()
                        a, b = cur[s][0][off:off + k], base[s][0][off:off + k]
                        if not torch.equal(a, b):
                            which.append(f"{n} ({int((a != b).sum())} of {k}, max {float((a - b).abs().max()):.1e})")
                        off += k
                hk = [k for k in cur[s][2] if not torch.equal(cur[s][2][k], base[s][2][k])]
                if cur[s][2]:
                    print(f"    hooked tensors that differ: {hk}")
                    for k in hk:
                        a, b = cur[s][2][k], base[s][2][k]
                        if a.dim() == 1:
                            idx = (a != b).nonzero().flatten()[:6].tolist()
                            print(f"      {k}: rows {idx}: now {[float(a[i]) for i in idx]} base {[float(b[i]) for i in idx]}")
                        else:
                            rows = (a != b).any(1).nonzero().flatten()
                            cols = (a != b).any(0).nonzero().flatten()
                            print(f"      {k}: {int((a != b).sum())} elements in {rows.numel()} rows {rows[:8].tolist()} and "
                                  f"{cols.numel()} columns {cols[:6].tolist()}..{cols[-3:].tolist()}, max |diff| {float((a - b).abs().max()):.2e}")
                print(f"rep {rep} step {s}: gradient bucket differs: {gd} [{'; '.join(which[:8])}] | parameters differ: {pd[:6]}", flush=True)
                break
    print(f"{bad} of {reps} repetitions differ", flush=True)


if __name__ == "__main__":
    main()
