#!/usr/bin/env python3
"""Does an RCCL process group in the process - its streams, its collective between the replayed step and Adam - push the
two-stream captured step off its hardware-queue configuration (profiles/r05/r_hw_queue_sweep.txt: a 40 % cliff)?
One rank, world size 1, the N > 1 form of bench.py's loop: replay(fwd + bwd graph) -> all_reduce(AVG) on the flat gradient
bucket -> dc_adam_flat, eagerly.  MODE = none (no process group, eager Adam tail) | pg (group created, no collective) |
ar (collective every step) | ar_graph (collective + Adam captured with the step); suffix _tl: the capture in
capture_error_mode="thread_local" (the process group's watchdog thread queries events while this thread captures)."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(mode):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if mode != "none":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deformcontact_amd import dp, synth
    from deformcontact_amd import graph as dc_graph
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = (b.to(dev) for b in synth.make_batch(32))
    rest.assume_segments(rest.segments())
    rig.assume_segments(rig.segments())
    pool = [(rest.edge_index.clone(), rig.edge_index.clone())]
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    g_s = torch.randn(rest.x.shape[0], 256, device=dev)
    g_r = torch.randn(rig.x.shape[0], 256, device=dev)
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    bucket.zero()

    def fwd_bwd():
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [g_s, g_r])

    def tail():
        bucket.wait_direct_writes()
        if mode.startswith("ar"):
            dist.all_reduce(bucket.flat, op=dist.ReduceOp.AVG)
        opt.step()

    def new_batch():
        rest.edge_index.copy_(pool[0][0])          # a "new" edge_index every step: the adjacency is rebuilt inside the graph
        rig.edge_index.copy_(pool[0][1])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
            tail()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    dc_graph.clear_cache()
    gr = torch.cuda.CUDAGraph()
    kw = {"capture_error_mode": "thread_local"} if mode.endswith("_tl") else {}
    with torch.cuda.graph(gr, **kw):
        fwd_bwd()
        if mode.startswith("ar_graph"):
            tail()

    def step():
        new_batch()
        gr.replay()
        if not mode.startswith("ar_graph"):
            tail()

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e3)
    print(f"{mode:9s} ms per step: " + " ".join(f"{r:.4f}" for r in res), flush=True)
    if mode != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
