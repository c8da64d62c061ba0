#!/usr/bin/env python3
"""Hunt for the rare mismatch of tests/test_a_dp_graphed.py: ONE process, no collectives - the whole train step through
`train.GraphedTrainStep` (world size 1) against the same steps run eagerly, parameters compared after EVERY step, repeated;
prints the first step and the parameters that differ.   python tools/exp/dp_flake.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from deformcontact_amd.train import GraphedTrainStep, losses  # noqa: E402

STEPS, B = 5, 2


def batch(step, dev):
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=step * B, soft_vertices=256, sphere_resolution=8))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    bad = 0
    for rep in range(reps):
        torch.manual_seed(100)
        model = load_model(EVERYDAY_NETWORK).to(dev)
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        bucket = dp.GradBucket(model.parameters(), direct=True)
        opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
        bucket.zero()
        stepper = GraphedTrainStep(model, opt, bucket, 1.0, eager_steps=1)
        snaps = []
        for s in range(STEPS):
            stepper(*batch(s, dev))
            snaps.append({n: p.detach().clone() for n, p in model.named_parameters()})
        ref = load_model(EVERYDAY_NETWORK).to(dev)
        ref.load_state_dict(init)
        rb = dp.GradBucket(ref.parameters(), direct=True)
        ro = dp.FlatAdam(rb, lr=4e-4, zero_grad_in_step=True)
        rb.zero()
        first = None
        for s in range(STEPS):
            losses(ref, *batch(s, dev), 1.0)["loss"].backward()
            rb.wait_direct_writes()
            ro.step()
            torch.cuda.synchronize()
            diff = [(n, float((p.detach() - snaps[s][n]).abs().max())) for n, p in ref.named_parameters()
                    if not torch.equal(p.detach(), snaps[s][n])]
            if diff and first is None:
                first = (s, diff)
        if first is not None:
            bad += 1
            print(f"rep {rep}: first mismatch after step {first[0]} (step 0 is eager in both): "
                  + ", ".join(f"{n} {d:.2e}" for n, d in first[1][:12]) + (f" ... {len(first[1])} tensors" if len(first[1]) > 12 else ""),
                  flush=True)
    print(f"{bad} of {reps} repetitions differ")


if __name__ == "__main__":
    main()
