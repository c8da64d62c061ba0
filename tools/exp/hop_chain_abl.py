#!/usr/bin/env python3
"""Timing-only ablations of dc_hop_chain_f32 (DC_CHAIN_ABL bits, see dc_hopchain.hip) and a K sweep, in the step regime
(four rotating slabs, graph-replayed, HIP events): where the 3-hop chain launch spends its time."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, current_stream_ptr  # noqa: E402

F = 256
GCN = os.environ.get("GCN", "1") != "0"          # the LDS-adjacency kernel (default) or the id / weight loading one
CS = os.path.join(ROOT, "deformcontact_amd", "csrc")


def build(bits):
    so = os.path.join(HERE, f"libchain_abl_{bits}.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-Wno-unused-value", f"-DDC_CHAIN_ABL={bits}", os.path.join(CS, "dc_hopchain.hip"),
                           os.path.join(CS, "dc_core.hip"), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.dc_hop_chain_f32.argtypes = [vp, vp, vp, vp, i64, ctypes.POINTER(i64), ci, vp, i64, i64, i64, ci, ci, ci, vp, ci, vp]
    return X


def chain(X, g, slab, rm, bwd, k):
    adj = g.bwd if bwd else g.fwd
    nptr, _, nseg = g._segments
    rc = X.dc_hop_chain_f32(adj.ptr.data_ptr(), adj.other.data_ptr(), adj.w.data_ptr(), g.fwd.ptr.data_ptr() if GCN else None, adj.other.numel(), nptr, nseg,
                            slab.data_ptr(), slab.stride(0), slab.size(0), F, k, 0, 1, rm.data_ptr() if rm is not None else None,
                            2 if bwd else 1, current_stream_ptr(slab.device))
    assert rc == 0


def timed(fn, reps=20, inner=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(inner):
            fn()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3


def main():
    dev = torch.device("cuda:0")
    masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 32, 4, 8, 16, 1 + 2, 1 + 8, 1 + 2 + 8, 1 + 2 + 4 + 8 + 16]
    rest, _, rig = synth.make_batch(32)
    graphs = [GraphIndex(b.edge_index.to(dev), b.x.shape[0], segments=b.segments()) for b in (rest, rig)]
    seq = [(g, ops._alloc_slab(g.num_nodes, 4 * F, dev).normal_(), torch.zeros(g.num_nodes, device=dev), bwd)
           for g in graphs for bwd in (False, True)]
    for bits in masks:
        X = build(bits)
        line = f"ABL {bits:3d}:"
        for k in (3, 2, 1):
            us = timed(lambda: [chain(X, g, s, r, b, k) for g, s, r, b in seq])
            line += f"  K={k}: {us:6.1f} us/step"
        s0 = timed(lambda: chain(X, *seq[0], 3))
        r0 = timed(lambda: chain(X, *seq[2], 3))
        s0n = timed(lambda: chain(X, seq[0][0], seq[0][1], None, False, 3))
        print(line + f"   soft fwd alone {s0:5.1f} (no rowmax {s0n:5.1f})  rigid fwd alone {r0:5.1f}", flush=True)


if __name__ == "__main__":
    main()
