#!/usr/bin/env python3
"""Determinism / stability soak (GPU box): two identical runs of N graph-replayed encoder steps
(fwd + bwd + flat Adam) must end with bit-identical parameters, all finite.

    python tools/soak.py [--steps 2000]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402


def run(steps, dev):
    rest, _, rig = synth.make_batch(8)
    rest, rig = rest.to(dev), rig.to(dev)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    g_rest = torch.randn(rest.x.shape[0], 256, device=dev, generator=gen) * 1e-3
    g_rig = torch.randn(rig.x.shape[0], 256, device=dev, generator=gen) * 1e-3
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=1e-4, zero_grad_in_step=True)
    bucket.zero()

    def one():
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [g_rest, g_rig])
        opt.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    return opt.flat_param.clone(), float(opt.step_count[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    p1, n1 = run(a.steps, dev)
    p2, n2 = run(a.steps, dev)
    ok = bool(torch.isfinite(p1).all()) and torch.equal(p1, p2) and n1 == n2 == a.steps + 3
    print(f"steps {a.steps}: finite={bool(torch.isfinite(p1).all())} identical={torch.equal(p1, p2)} "
          f"adam_steps={n1}/{n2} |p|max={float(p1.abs().max()):.4f} -> {'OK' if ok else 'FAIL'}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
