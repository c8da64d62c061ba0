"""Determinism soak of the eager whole-model train step with STOCK attention kernels on two encoder streams - the
harness in which a chain workgroup that left LDS free on its compute unit returned a few wrong elements in 0.2 - 1.75 % of
repetitions (rounds 4 - 6: profiles/r05/README.md, profiles/r06/README.md).  The shipped chain kernels own the whole LDS of their
CU (`dc_hopchain.hip: kChainLdsRequest`); this keeps the reproducer in the GPU suite: every repetition's gradient
bucket must equal the first one's bit for bit.  `DC_SOAK_REPS` (default 200, ~35 s; the hunts ran 1,500 - 3,000).
Reference: /root/reference/train.py:46-58,71-73 (the step), models/model.py:13-21 (the attention whose stock kernels share the CUs)."""
import os

import pytest
import torch

from deformcontact_amd import dp, synth
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, CrossAttention, load_model
from deformcontact_amd.train import losses

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS, B = 4, 2


def _run(init, batches):
    m = load_model(EVERYDAY_NETWORK).to(DEV)
    m.load_state_dict(init)
    bk = dp.GradBucket(m.parameters(), direct=True)
    op = dp.FlatAdam(bk, lr=4e-4, zero_grad_in_step=True)
    bk.zero()
    snaps = []
    for s in range(STEPS):
        losses(m, *[b.clone() for b in batches[s]], 1.0)["loss"].backward()
        bk.wait_direct_writes()
        snaps.append(bk.flat.clone())
        op.step()
        torch.cuda.synchronize()
    return snaps


def test_two_stream_train_step_with_stock_attention_is_bit_reproducible(monkeypatch):
    monkeypatch.setattr(CrossAttention, "fused", "0")            # stock rocBLAS / ATen attention kernels in the step
    reps = int(os.environ.get("DC_SOAK_REPS", "200"))
    batches = [tuple(b.to(DEV) for b in synth.make_batch(B, first_idx=s * B, soft_vertices=256, sphere_resolution=8))
               for s in range(STEPS)]
    torch.manual_seed(100)
    init = {k: v.detach().clone() for k, v in load_model(EVERYDAY_NETWORK).to(DEV).state_dict().items()}
    torch.cuda.synchronize()
    base = _run(init, batches)
    bad = []
    for rep in range(reps):
        cur = _run(init, batches)
        for s in range(STEPS):
            if not torch.equal(cur[s], base[s]):
                bad.append((rep, s, int((cur[s] != base[s]).sum())))
                break
    assert not bad, f"{len(bad)} of {reps} repetitions differ from the first: {bad[:5]}"
