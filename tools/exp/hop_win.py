#!/usr/bin/env python3
"""Drive tools/exp/hop_win.hip (the LDS-window hop, an r03 experiment): bit-identity against the product's gather hop on
the B=32 batch, then the F=256 hop chains of a step (3 forward + 3 transposed hops per adjacency, row maxima) timed
graph-replayed in the step regime with both kernels, per-branch and over the merged adjacency."""
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


def load():
    so = os.path.join(HERE, "libhopwin.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "hop_win.hip"), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.hop_win_run.argtypes = [vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i64, vp, ci, vp]
    return X


def win_hop(X, adj, x, out, rowmax, mode):
    rc = X.hop_win_run(adj.ptr.data_ptr(), adj.other.data_ptr(), adj.w.data_ptr(), x.data_ptr(), x.stride(0), None, 0,
                       out.data_ptr(), out.stride(0), x.shape[0], F, rowmax.data_ptr(), mode,
                       current_stream_ptr(x.device))
    assert rc == 0


def chain(X, g, slab, rm, bwd, window):
    adj = g.bwd if bwd else g.fwd
    for j in range(3):
        mode = 2 if (bwd or j > 0) else 1
        x, o = slab[:, j * F:(j + 1) * F], slab[:, (j + 1) * F:(j + 2) * F]
        if window:
            win_hop(X, adj, x, o, rm, mode)
        else:
            ops.hop(adj, x, out=o, rowmax=rm, rowmax_mode=mode)


def chain_time(X, graphs, window, reps=20):
    dev = torch.device("cuda:0")
    seq = [(g, ops._alloc_slab(g.num_nodes, 4 * F, dev).normal_(), torch.zeros(g.num_nodes, device=dev), bwd)
           for g in graphs for bwd in (False, True)]

    def fn():
        for g, slab, rm, bwd in seq:
            chain(X, g, slab, rm, bwd, window)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(5):
            fn()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * 5) * 1e3        # us per step's worth of F=256 hops


def main():
    X = load()
    dev = torch.device("cuda:0")
    rest, _, rig = synth.make_batch(32)
    parts = [(rest.edge_index.to(dev), rest.x.shape[0]), (rig.edge_index.to(dev), rig.x.shape[0])]
    for ei, n in parts:                                  # same bits as the product hop, row maxima included
        g = GraphIndex(ei, n)
        a, b = ops._alloc_slab(n, 4 * F, dev), None
        a[:, :F].normal_()
        b = a.clone()
        ra, rb = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        for bwd in (False, True):
            chain(X, g, a, ra, bwd, False)
            chain(X, g, b, rb, bwd, True)
            torch.cuda.synchronize()
            assert torch.equal(a, b) and torch.equal(ra, rb), "window hop differs from the product hop"
    n = sum(p[1] for p in parts)
    e = sum(int(p[0].shape[1]) for p in parts)
    comp = 6 * (e * 8 + n * (8 * F + 4))
    for name, mk in (("per-branch (12 launches)", lambda: [GraphIndex(ei, nn) for ei, nn in parts]),
                     ("merged (6 launches)", lambda: [GraphIndex.from_parts(parts)])):
        for window in (False, True):
            us = chain_time(X, mk(), window)
            print(f"{name:26s} {'LDS window' if window else 'gather    '}: {us:7.1f} us per step's F=256 hops, "
                  f"compulsory-bytes frac of 8 TB/s {comp / us / 1e6 / 8.0:.3f}")


if __name__ == "__main__":
    main()
