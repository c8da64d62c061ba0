"""Worker of tests/launch_scenarios.py (dp_graphed_*; asserted on by tests/test_z_launch.py): one data-parallel rank (env RANK / WORLD_SIZE / MASTER_*), every rank on
cuda:0, collectives over gloo.  The reference's whole train step through `train.GraphedTrainStep` with
world_size 2: forward + losses + backward replayed from one hipGraph, gradient all-reduce + Adam OUTSIDE the
graph.  Rank 0 then replays the same steps in ONE process - both ranks' batches one after the other, mean of the
two gradient sets, the same Adam kernel - and asserts that the data-parallel replicas hold exactly those
parameters.  (Not a pytest file: started as a child process before the parent touches the GPU.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

STEPS, B = 5, 2


def batch(step, rank, world, dev):
    from deformcontact_amd import synth
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=(step * world + rank) * B, soft_vertices=256,
                                                     sphere_resolution=8))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")            # one node: pairwise connections over loopback
    import datetime
    from deformcontact_amd.launch import install_watchdog, phase
    install_watchdog()
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "4")))      # ranks share the node's cores (launch.py)
    phase("start")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(
        seconds=float(os.environ.get("DC_GLOO_TIMEOUT_S", "60"))))
    phase("process group up")
    from deformcontact_amd import dp
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, ContactEncoder, load_model
    from deformcontact_amd.train import GraphedTrainStep, losses
    if os.environ.get("DC_TEST_SERIAL_BRANCHES") == "1":
        ContactEncoder.overlap_branches = False        # both encoder branches on the caller's stream
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + rank)                       # different init per rank ...
    model = load_model(EVERYDAY_NETWORK).to(dev)
    dp.broadcast_parameters(model)                      # ... made identical here
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bucket = dp.GradBucket(model.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    bucket.zero()
    stepper = GraphedTrainStep(model, opt, bucket, 1.0, eager_steps=1)
    assert not stepper._tail_in_graph and dist.get_world_size() == world
    loss_curve = []
    for s in range(STEPS):
        loss_curve.append(float(stepper(*batch(s, rank, world, dev))["loss"]))
        phase(f"step {s} done")
    torch.cuda.synchronize()
    assert stepper.replays == STEPS - 1
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # every replica holds the same parameters
    mine_c = mine.cpu()
    gathered = [torch.empty_like(mine_c) for _ in range(world)]
    dist.all_gather(gathered, mine_c)
    assert all(torch.equal(g, mine_c) for g in gathered), "replicas diverged"
    result = {"rank": rank, "losses": loss_curve, "replays": stepper.replays}
    if rank == 0:
        # single-process reference: the mean over the ranks' gradients, step by step
        torch.manual_seed(0)
        ref = load_model(EVERYDAY_NETWORK).to(dev)
        ref.load_state_dict(init)
        rb = dp.GradBucket(ref.parameters(), direct=True)
        ro = dp.FlatAdam(rb, lr=4e-4, zero_grad_in_step=True)
        rb.zero()
        for s in range(STEPS):
            acc = torch.zeros_like(rb.flat)
            for r in range(world):
                rb.zero()
                losses(ref, *batch(s, r, world, dev), 1.0)["loss"].backward()
                rb.wait_direct_writes()
                acc += rb.flat
            rb.flat.copy_(acc)                           # gloo: SUM over ranks ...
            rb.flat.div_(world)                          # ... then / world (dp.GradBucket.all_reduce_mean)
            ro.step()
        torch.cuda.synchronize()
        want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
        diff = float((mine - want).abs().max())
        scale = float(want.abs().max())
        differing = [n for (n, p), q in zip(model.named_parameters(), ref.parameters()) if not torch.equal(p.detach(), q.detach())]
        result.update(max_abs_diff=diff, scale=scale, bit_identical=bool(torch.equal(mine, want)), differing=differing[:12],
                      n_differing=len(differing))
    phase("compared")
    dist.barrier()
    with open(out_path + f".rank{rank}.json", "w") as f:
        json.dump(result, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
