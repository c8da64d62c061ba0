#!/usr/bin/env python3
"""Repeat the flash attention forward + backward at the batch-32 shape and at a ragged shape and demand BIT-identical
results every time: the kernels synchronise LDS-DMA, fragment reads and MFMAs with hand-written wait counts - a missing
wait shows up as run-to-run differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import attention  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    for ns, nr in ((32768, 24384), (5000, 3100), (333, 95)):
        torch.manual_seed(ns)
        q = (torch.randn(ns, 256, device=dev) * 0.3).requires_grad_()
        k = (torch.randn(nr, 256, device=dev) * 0.3).requires_grad_()
        v = torch.randn(nr, 256, device=dev).requires_grad_()
        go = torch.randn(ns, 256, device=dev)
        ref = None
        bad = 0
        for it in range(reps):
            q.grad = k.grad = v.grad = None
            o = attention.attention_core(q, k, v)
            o.backward(go)
            cur = [t.detach().clone() for t in (o, q.grad, k.grad, v.grad)]
            if ref is None:
                ref = cur
            else:
                bad += sum(int(not torch.equal(a, b)) for a, b in zip(ref, cur))
        finite = all(bool(torch.isfinite(t).all()) for t in ref)
        print(f"ns={ns} nr={nr}: {reps} repetitions, mismatching tensors {bad}, finite {finite}")
        assert bad == 0 and finite


if __name__ == "__main__":
    main()
