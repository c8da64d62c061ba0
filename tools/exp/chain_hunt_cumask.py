#!/usr/bin/env python3
"""Round 5: does the chain launch's rare run-to-run difference need the two encoder branches to SHARE compute units?
The same loop as chain_hunt.py, with the caller's stream and the encoder's side stream created by
hipExtStreamCreateWithCUMask: HUNT_CUMASK=disjoint - the two streams own disjoint halves of the CUs (kernels of the two
branches still run at the same time, never on the same CU); HUNT_CUMASK=same - both streams carry the full mask (control:
same stream objects, CUs shared as always).   python tools/exp/chain_hunt_cumask.py [repeats]"""
import ctypes
import os
import sys

os.environ.setdefault("DC_HOP_CHAIN_GCN_MIN_NODES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import dp, ops, synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, ContactEncoder, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402

assert getattr(ops, "HOP_CHAIN_GCN_MIN_NODES", 0) == 0, "DC_HOP_CHAIN_GCN_MIN_NODES must be 0 in the environment"
STEPS, B = 4, 2
MODE = os.environ.get("HUNT_CUMASK", "disjoint")


def masked_stream(words):
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)


def batch(step, dev):
    return tuple(b.to(dev) for b in synth.make_batch(B, first_idx=step * B, soft_vertices=256, sphere_resolution=8))


def run(init, dev):
    m = load_model(EVERYDAY_NETWORK).to(dev)
    m.load_state_dict(init)
    bk = dp.GradBucket(m.parameters(), direct=True)
    op = dp.FlatAdam(bk, lr=4e-4, zero_grad_in_step=True)
    bk.zero()
    snaps = []
    for s in range(STEPS):
        losses(m, *batch(s, dev), 1.0)["loss"].backward()
        bk.wait_direct_writes()
        grads = bk.flat.clone()
        op.step()
        torch.cuda.synchronize()
        snaps.append(grads)
    return snaps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    torch.cuda.init()
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    nw = (ncu + 31) // 32
    full = [0xFFFFFFFF] * nw
    if MODE == "disjoint":
        # alternate 32-CU words: whatever the bit -> (XCD, CU) mapping is, the two sets are disjoint
        a = [0xFFFFFFFF if i % 2 == 0 else 0 for i in range(nw)]
        b = [0 if i % 2 == 0 else 0xFFFFFFFF for i in range(nw)]
    else:
        a, b = full, full
    main_s, side_s = masked_stream(a), masked_stream(b)
    ContactEncoder._side_streams[(dev.type, dev.index)] = side_s
    print(f"{ncu} CUs, masks main {['%08x' % w for w in a]} side {['%08x' % w for w in b]}", flush=True)
    torch.manual_seed(100)
    init = {k: v.detach().clone() for k, v in load_model(EVERYDAY_NETWORK).to(dev).state_dict().items()}
    torch.cuda.synchronize()
    bad = nans = 0
    with torch.cuda.stream(main_s):
        base = run(init, dev)
        for rep in range(reps):
            cur = run(init, dev)
            for s in range(STEPS):
                if not torch.equal(cur[s], base[s]):
                    bad += 1
                    # (round 6: with a -DDC_CHAIN_POISON build a NaN here = LDS read before its data had landed)
                    n_nan = int(torch.isnan(cur[s]).sum())
                    nans += n_nan > 0
                    d = (cur[s] - base[s]).abs()
                    print(f"rep {rep} step {s}: gradient bucket differs ({int((d > 0).sum())} elements, max |d| "
                          f"{float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else float('nan'):.3e}, NaNs {n_nan})",
                          flush=True)
                    break
    print(f"HUNT_CUMASK={MODE}: {bad} of {reps} repetitions differ from the first ({nans} of them with NaNs; "
          f"base has NaNs: {any(bool(torch.isnan(b).any()) for b in base)})", flush=True)


if __name__ == "__main__":
    main()
