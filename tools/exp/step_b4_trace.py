#!/usr/bin/env python3
"""Whole reference train step at the shipped batch size 4, eager, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/exp/step_b4_trace.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    rest, deff, rig = (b.to(dev) for b in synth.make_batch(batch))
    torch.manual_seed(0)
    model = load_model(EVERYDAY_NETWORK).to(dev)
    bucket = dp.GradBucket(model.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    bucket.zero()
    for _ in range(30):
        o = losses(model, rest, deff, rig, 1.0)
        o["loss"].backward()
        opt.step()
    torch.cuda.synchronize()
    print("done", float(o["loss"]))


if __name__ == "__main__":
    main()
