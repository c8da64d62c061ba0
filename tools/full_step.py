#!/usr/bin/env python3
"""Whole reference train step (train.py:46-58,71-73) at a given batch size: encoder on the HIP path,
unmasked cross-attention + decoder + losses on stock PyTorch.  python tools/full_step.py --batch 32"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from deformcontact_amd.train import train_step  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--graph", action="store_true",
                    help="capture losses + backward + FlatAdam in one hipGraph (flat-bucket Adam kernel)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rest, deff, rig = (b.to(dev) for b in synth.make_batch(a.batch))
    torch.manual_seed(0)
    model = load_model(EVERYDAY_NETWORK).to(dev)
    if a.graph:
        from deformcontact_amd import dp
        from deformcontact_amd.train import losses
        bucket = dp.GradBucket(model.parameters(), direct=True)         # as bench.py and train.train: dW straight into the bucket
        opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
        bucket.zero()
        out = {}

        def one():
            o = losses(model, rest, deff, rig, 1.0)
            o["loss"].backward()
            opt.step()
            return o["loss"].detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                one()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss_t = one()
        run = g.replay
    else:
        opt = torch.optim.Adam(model.parameters(), lr=4e-4)

        def run():
            out.update(train_step(model, opt, rest, deff, rig))
        out = {}
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    if a.graph:
        out = {"loss": loss_t}
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    e = rest.edge_index.shape[1] + rig.edge_index.shape[1]
    ns, nr = rest.x.shape[0], rig.x.shape[0]
    att_flop = 2 * 3 * (4.0 * ns * nr * 256)          # 2 heads, fwd + 2x bwd
    print(f"B={a.batch}: {ms:.2f} ms/step, {e / ms / 1e3:.2f} M edges/s, loss {float(out['loss']):.6f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, attention ~{att_flop / 1e12:.2f} TFLOP "
          f"=> >= {att_flop / 155e12 * 1e3:.1f} ms at the fp32 MFMA peak")


if __name__ == "__main__":
    main()
