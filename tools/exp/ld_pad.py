#!/usr/bin/env python3
"""Does the power-of-two row stride of the hop slab (1024 floats = 4 KiB) cost bandwidth?  The wide
forward block (k_fwd_h2, weights pre-split) and the F=256 hop over the same data with the slab's leading
dimension padded by 0 / 16 / 32 / 64 / 80 floats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _ptr_array  # noqa: E402


def gtime(fn, calls=10, reps=8):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * calls)


dev = torch.device("cuda:0")
L = _lib.lib()
rest, _, _ = synth.make_batch(32)
n, f, fo = rest.x.shape[0], 256, 256
gi = GraphIndex(rest.edge_index.to(dev), n)
ws = [torch.randn(fo, f, device=dev) / 16 for _ in range(4)]
wmax = torch.empty(fo, device=dev)
wimg = torch.empty(fo, 4 * f, device=dev)
bias = torch.randn(fo, device=dev)
out = torch.empty(n, fo, device=dev)
for pad in (0, 16, 32, 64, 80, 256):
    ld = 4 * f + pad
    # several slabs in rotation so that the operands come from beyond the Infinity Cache, as in a step
    bufs = [torch.randn(n, ld, device=dev) for _ in range(3)]
    rms = [b[:, :4 * f].abs().amax(1).contiguous() for b in bufs]
    state = {"i": 0}

    def fwd():
        st = current_stream_ptr(dev)
        i = state["i"] = (state["i"] + 1) % 3
        L.dc_tag_weight_prep(_ptr_array(ws), 4, fo, f, wmax.data_ptr(), wimg.data_ptr(), None, None, st)
        L.dc_tag_linear_fwd_h2p(bufs[i].data_ptr(), ld, wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n,
                                4 * f, fo, rms[i].data_ptr(), wmax.data_ptr(), None, 0, st)

    def hops():
        i = state["i"] = (state["i"] + 1) % 3
        for j in range(3):
            ops.hop(gi.fwd, bufs[i][:, j * f:(j + 1) * f], out=bufs[i][:, (j + 1) * f:(j + 2) * f])
    tf = gtime(fwd)
    th = gtime(hops) / 3
    print(f"slab ld = {ld:5d} floats ({ld * 4} B): weight_prep + k_fwd_h2 {tf * 1e3:7.1f} us   hop {th * 1e3:6.2f} us")
