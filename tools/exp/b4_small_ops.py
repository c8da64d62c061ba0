#!/usr/bin/env python3
"""Which stock PyTorch elementwise kernels does a batch-4 train step launch, and from where?  (profiles: 42 `add` launches
per step, 10 % of the step.)   python tools/exp/b4_small_ops.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rest, deff, rig = (b.to(dev) for b in synth.make_batch(4))
    torch.manual_seed(0)
    model = load_model(EVERYDAY_NETWORK).to(dev)
    bucket = dp.GradBucket(model.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    bucket.zero()

    def one():
        losses(model, rest, deff, rig, 1.0)["loss"].backward()
        opt.step()
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        one()
        torch.cuda.synchronize()
    by = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::copy_", "aten::zeros", "aten::fill_",
                       "aten::zero_", "aten::cat", "aten::mean", "aten::sub", "aten::div", "aten::clone", "aten::contiguous"):
            frames = [f for f in (ev.stack or []) if "deformcontact_amd" in f or "tools/" in f]
            where = frames[0].split("deformcontact_amd/")[-1] if frames else "(autograd engine / no python frame)"
            shapes = str(ev.input_shapes)[:60]
            by[(ev.name, where[:70], shapes)] += 1
    for (name, where, shapes), n in sorted(by.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{n:3d} x {name:16s} {where:72s} {shapes}")


if __name__ == "__main__":
    main()
