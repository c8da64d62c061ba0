#!/usr/bin/env python3
"""Drive tools/exp/hop_union.hip (the bf16 hop with a run's neighbour rows staged in LDS once) on the Morton-ordered
100k-point radius graph of BASELINE configs[4]: list build time, bit-identity of the hop against `ops.hop_bf16`
(dc_spmm_bf16: same products, same order), both timed graph-replayed.  Round 4's last run: bit-identical, 93 - 98 us
against 53 - 55 us (DESIGN.md section 8 says why and what it would take).

    python tools/exp/hop_union.py
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, NodeOrder, current_stream_ptr  # noqa: E402


def build():
    so = os.path.join(HERE, "libhopunion.so")
    src = os.path.join(HERE, "hop_union.hip")
    if not (os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src)):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", src, "-o", so])
    X = ctypes.CDLL(so)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    X.hu_build_lists.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    X.hu_hop_bf16.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, vp]
    return X


def timed(run, reps=20):
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    X = build()
    dev = torch.device("cuda:0")
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    pos, ei = pos.to(dev), ei.to(dev)
    n, f = pos.shape[0], 256
    order = NodeOrder.morton(pos)
    g = GraphIndex(order.relabel(ei), n)
    x = torch.randn(n, f, device=dev).bfloat16()
    runs = (n + X.hu_run_rows() - 1) // X.hu_run_rows()
    ulist = torch.empty(runs * X.hu_run_max(), dtype=torch.int32, device=dev)
    ucnt = torch.empty(runs, dtype=torch.int32, device=dev)
    lidx = torch.empty(g.num_edges, dtype=torch.int16, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for name, adj in (("forward", g.fwd), ("transposed", g.bwd)):
        def lists():
            X.hu_build_lists(adj.ptr.data_ptr(), adj.other.data_ptr(), n, ulist.data_ptr(), ucnt.data_ptr(), lidx.data_ptr(),
                             status.data_ptr(), current_stream_ptr(dev))
        lists()
        torch.cuda.synchronize()
        assert int(status) == 0, f"{name}: a run exceeds the caps (status {int(status)})"
        fetched = int(ucnt.sum())
        print(f"{name}: {runs} runs, {fetched} row fetches for {g.num_edges} edges ({g.num_edges / fetched:.2f} x fewer), "
              f"largest union {int(ucnt.max())}; lists built in {timed(lists):.1f} us")
        want = ops.hop_bf16(adj, x)
        got = torch.empty_like(want)

        def hop_u():
            X.hu_hop_bf16(adj.ptr.data_ptr(), adj.w.data_ptr(), lidx.data_ptr(), ulist.data_ptr(), ucnt.data_ptr(), x.data_ptr(),
                          x.stride(0), got.data_ptr(), got.stride(0), n, current_stream_ptr(dev))
        hop_u()
        torch.cuda.synchronize()
        same = torch.equal(got, want)
        print(f"{name}: bit-identical to dc_spmm_bf16: {same}" +
              ("" if same else f" (max |diff| {float((got.float() - want.float()).abs().max()):.3e})"))
        print(f"{name}: dc_spmm_bf16 {timed(lambda: ops.hop_bf16(adj, x, out=want)):.1f} us, staged-union hop {timed(hop_u):.1f} us")


if __name__ == "__main__":
    main()
