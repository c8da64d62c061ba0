#!/usr/bin/env python3
"""Which tensor of an encoder forward + backward is not reproducible?  Small meshes (the shapes of tests/dp_graphed_worker.py),
the same inputs every repetition, eager; outputs, input gradients of every layer (hooks) and every parameter gradient compared
bit for bit with the first repetition.   python tools/exp/det_hunt.py [repeats] [batch] [soft_vertices]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graph import clear_cache  # noqa: E402
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sv = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dev = torch.device("cuda:0")
    rest, _, rig = (b.to(dev) for b in synth.make_batch(bsz, soft_vertices=sv, sphere_resolution=8))
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    ga = torch.randn(rest.x.shape[0], 256, device=dev)
    gb = torch.randn(rig.x.shape[0], 256, device=dev)
    first, counts = None, {}
    noise = torch.cuda.Stream() if os.environ.get("HUNT_NOISE") == "1" else None
    gen = torch.Generator().manual_seed(7)
    for rep in range(reps):
        if noise is not None:
            # unrelated work on a third stream, different every repetition: other kernels' leftovers in LDS / registers and
            # a different interleaving with the encoder's two streams
            with torch.cuda.stream(noise):
                for _ in range(int(torch.randint(1, 6, (1,), generator=gen))):
                    m_ = int(torch.randint(64, 1500, (1,), generator=gen))
                    a_ = torch.randn(m_, 384, device=dev) * float(torch.rand(1, generator=gen) * 1e6)
                    (torch.softmax(a_ @ a_.t(), dim=-1) @ a_).sum()
        if rep % 3 == 0:
            clear_cache()
        enc.zero_grad(set_to_none=True)
        rec = {}
        hooks = []
        for name, layers in (("rest", enc.conv_layers_resting), ("rig", enc.conv_layers_rigid)):
            for i, l in enumerate(layers):
                hooks.append(l.register_forward_hook(lambda m, inp, out, k=f"{name}.{i}.out": rec.__setitem__(k, out.detach().clone())))
                hooks.append(l.register_full_backward_hook(
                    lambda m, gi, go, k=f"{name}.{i}": rec.__setitem__(k + ".grad_out", go[0].detach().clone())))
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [ga, gb])
        torch.cuda.synchronize()
        for h in hooks:
            h.remove()
        for n, p in enc.named_parameters():
            rec["param." + n] = p.grad.detach().clone()
        if first is None:
            first = rec
            continue
        for k, v in rec.items():
            if not torch.equal(v, first[k]):
                d = (v - first[k]).abs()
                counts.setdefault(k, []).append((rep, int((d > 0).sum()), float(d.max())))
    if not counts:
        print(f"{reps} repetitions at B={bsz}, {sv} soft vertices: everything bit-identical")
    for k, v in counts.items():
        print(f"{k}: differs in {len(v)} of {reps - 1} repetitions; e.g. rep {v[0][0]}: {v[0][1]} elements, max |diff| {v[0][2]:.3e}")


if __name__ == "__main__":
    main()
