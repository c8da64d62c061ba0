"""CPU, world_size 2 over gloo: the data-parallel path (sample sharding + one flat-bucket
gradient all-reduce) that runs over RCCL/xGMI on the GPU node.  The replicas use the oracle
convs on CPU tensors - this tests the DP wiring, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deformcontact_amd import dp, synth
from deformcontact_amd.graphnet import ContactEncoder
from oracle import pyg_ref


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(rank_seed):
    torch.manual_seed(rank_seed)
    return ContactEncoder([21, 25], 16, conv_module=pyg_ref)


def _local_grads(enc, rank):
    rest, _, rig = synth.make_batch(2, first_idx=2 * rank, soft_vertices=64, sphere_resolution=4)
    a, b = enc(rest, rig)
    gen = torch.Generator().manual_seed(100 + rank)
    ga = torch.randn(a.shape, generator=gen)
    gb = torch.randn(b.shape, generator=gen)
    torch.autograd.backward([a, b], [ga, gb])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        enc = _make(rank)                      # deliberately different init per rank
        dp.broadcast_parameters(enc, src=0)    # ... made identical here
        bucket = dp.GradBucket(enc.parameters())
        for step in range(2):                  # second step checks zero() + view re-use
            bucket.zero()
            _local_grads(enc, rank)
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket._views))
            bucket.all_reduce_mean()
        torch.save({"flat": bucket.flat.clone(),
                    "params": [p.detach().clone() for p in enc.parameters()]},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_serial_mean(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # replicas agree bit for bit after the collective, and parameters were broadcast from rank 0
    assert torch.equal(r0["flat"], r1["flat"])
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    # serial reference: same init (rank 0's), mean of the two ranks' gradients
    torch.set_num_threads(1)
    enc = _make(0)
    for a, b in zip(enc.parameters(), r0["params"]):
        assert torch.equal(a.detach(), b)
    grads = []
    for rank in range(world):
        enc.zero_grad(set_to_none=True)
        _local_grads(enc, rank)
        grads.append(torch.cat([p.grad.reshape(-1) for p in enc.parameters()]))
    ref = (grads[0] + grads[1]) / 2
    assert r0["flat"].numel() == ref.numel() == sum(p.numel() for p in enc.parameters())
    assert np.allclose(r0["flat"].numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)


def test_bucket_single_process_is_identity():
    enc = _make(3)
    bucket = dp.GradBucket(enc.parameters())
    bucket.zero()
    _local_grads(enc, 0)
    before = bucket.flat.clone()
    bucket.all_reduce_mean()                   # world size 1: no collective, no scaling
    assert torch.equal(before, bucket.flat) and float(before.abs().sum()) > 0
    # a replaced .grad is copied back into the bucket
    p0 = bucket.params[0]
    p0.grad = torch.ones_like(p0)
    bucket.all_reduce_mean()
    assert torch.equal(bucket._views[0], torch.ones_like(p0)) and p0.grad is bucket._views[0]
    with pytest.raises(ValueError):
        dp.GradBucket([])
