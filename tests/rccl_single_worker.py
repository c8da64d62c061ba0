"""One rank on cuda:0 over RCCL (backend "nccl"): the collectives bench.py / train.py issue at N > 1 - parameter
broadcast, the flat-bucket gradient all-reduce with ReduceOp.AVG, the float64 MAX of the timing, barrier - on the real
library, so that an unsupported op or dtype shows up on the 1-GPU test box and not first on the 8-GPU node."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deformcontact_amd import dp
    from deformcontact_amd.graphnet import ContactEncoder
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    dp.broadcast_parameters(enc)
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    bucket.zero()
    bucket.flat.fill_(3.0)
    dist.all_reduce(bucket.flat, op=dist.ReduceOp.AVG)            # what GradBucket.all_reduce_mean issues at N > 1
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # bench.py's max-over-ranks of the elapsed time
    dist.barrier()
    seen = [None]
    dist.all_gather_object(seen, {"rank": 0, "device": str(dev)})   # bench.py's `dist.ranks_seen`
    assert seen == [{"rank": 0, "device": str(dev)}]
    torch.cuda.synchronize()
    assert float(bucket.flat.min()) == 3.0 and float(bucket.flat.max()) == 3.0 and float(t) == 1.25
    assert dist.get_backend() == "nccl" and bucket.flat.numel() * 4 >= 572416 * 4
    # a hipGraph capture while the process group's watchdog thread polls the events of collectives issued just before
    # (dp.capture: thread-local capture checks - in the default global mode that poll is an error which aborts the
    # process, whenever it falls inside the capture: here the capture is held open long enough for several polls),
    # with the collective itself captured and replayed
    import time
    for _ in range(4):
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with dp.capture(gr):
        bucket.flat.mul_(2.0)
        time.sleep(1.0)
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.AVG)
        bucket.flat.add_(1.0)
    gr.replay()
    gr.replay()
    torch.cuda.synchronize()
    assert float(bucket.flat.min()) == 15.0 and float(bucket.flat.max()) == 15.0      # ((3 * 2 + 1) * 2 + 1)
    dist.destroy_process_group()
    print("RCCL_OK", bucket.flat.numel())


if __name__ == "__main__":
    main()
