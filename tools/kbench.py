#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the C-ABI entry points at the BASELINE shapes (GPU box).

    python tools/kbench.py [--reps 50]

Interleaves the variants in one process and reports median time, TF/s (dense) or GB/s of
algorithmic bytes (hop)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _i64_array, _ptr_array  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--products", type=int, default=6, help="bf16 products per tile for the *6 rows")
    args = ap.parse_args()
    global NP
    NP = args.products
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = current_stream_ptr(dev)
    for name, n, fi, nseg in (("soft L2", 32768, 256, 4), ("rigid L2", 24384, 256, 4),
                              ("soft L1 pad", 32768, 96, 1), ("rigid L1 pad", 24384, 112, 1)):
        fo = 256
        slab = torch.randn(n, nseg * fi, device=dev)
        xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
        ld = [nseg * fi] * nseg
        ws = [torch.randn(fo, fi, device=dev) / fi ** 0.5 for _ in range(nseg)]
        bias = torch.randn(fo, device=dev)
        out = torch.empty(n, fo, device=dev)
        g = torch.randn(n, fo, device=dev)
        gws = [torch.empty(fo, fi, device=dev) for _ in range(nseg)]
        gb = torch.empty(fo, device=dev)
        nbytes = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        gslab = torch.empty(n, nseg * fi, device=dev)
        gxs = [gslab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
        pa_x, pa_w, pa_gw, pa_gx, pa_ld = _ptr_array(xs), _ptr_array(ws), _ptr_array(gws), _ptr_array(gxs), _i64_array(ld)
        flops = 2.0 * n * fi * nseg * fo

        def fwd():
            L.dc_tag_linear_fwd(pa_x, pa_ld, pa_w, nseg, bias.data_ptr(), 1, out.data_ptr(), fo, n, fi, fo, st)

        def dx():
            L.dc_tag_linear_bwd_dx(g.data_ptr(), fo, out.data_ptr(), fo, pa_w, nseg, pa_gx, pa_ld, n, fi, fo, st)

        def dw():
            L.dc_tag_linear_bwd_dw(g.data_ptr(), fo, out.data_ptr(), fo, pa_x, pa_ld, nseg, pa_gw, nseg, fi, gb.data_ptr(), 0,
                                   scratch.data_ptr(), nbytes, n, fi, fo, st)

        wsb = L.dc_tag_linear_bwd_dx_split_workspace_bytes(fi, fo, nseg)
        wsx = torch.empty(wsb, dtype=torch.uint8, device=dev)

        def fwd_s():
            L.dc_tag_linear_fwd_split(pa_x, pa_ld, pa_w, nseg, bias.data_ptr(), 1, out.data_ptr(), fo, n, fi, fo, NP, st)

        def dx_s():
            L.dc_tag_linear_bwd_dx_split(g.data_ptr(), fo, out.data_ptr(), fo, pa_w, nseg, pa_gx, pa_ld,
                                         wsx.data_ptr(), wsb, n, fi, fo, NP, st)

        def dw_s():
            L.dc_tag_linear_bwd_dw_split(g.data_ptr(), fo, out.data_ptr(), fo, pa_x, pa_ld, nseg, pa_gw, nseg, fi,
                                         gb.data_ptr(), 0, scratch.data_ptr(), nbytes, n, fi, fo, NP, st)

        rowmax = slab.abs().amax(1).contiguous()
        growmax = torch.empty(n, device=dev)
        wmax = ops.weight_rowmax(ws)

        wimg = torch.empty(fo, nseg * fi, device=dev)
        if (nseg * fi) % 16 == 0:
            L.dc_tag_weight_prep(pa_w, nseg, fo, fi, wmax.data_ptr(), wimg.data_ptr(), None, None, st)

        def fwd_h():
            L.dc_tag_linear_fwd_h2p(slab.data_ptr(), nseg * fi, wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(),
                                    fo, n, nseg * fi, fo, rowmax.data_ptr(), wmax.data_ptr(), None, 0, st)

        def dx_h():
            L.dc_tag_linear_bwd_dx_h2(g.data_ptr(), fo, out.data_ptr(), fo, pa_w, nseg, pa_gx, pa_ld,
                                      wsx.data_ptr(), wsb, n, fi, fo, growmax.data_ptr(), wmax.data_ptr(), st)

        def dw_h():
            L.dc_tag_linear_bwd_dw_h2(g.data_ptr(), fo, None, fo, pa_x, pa_ld, nseg, pa_gw, nseg, fi,
                                      gb.data_ptr(), 0, scratch.data_ptr(), nbytes, n, fi, fo,
                                      growmax.data_ptr(), rowmax.data_ptr(), st)

        def amax():
            L.dc_rowabsmax_f32(g.data_ptr(), fo, n, fo, growmax.data_ptr(), st)
            L.dc_tag_weight_rowmax(pa_w, nseg, fo, fi, wmax.data_ptr(), st)

        amax()

        h2 = (("fwdH", fwd_h), ("dXH", dx_h), ("dWH", dw_h), ("amax", amax)) if fi % 16 == 0 else ()
        for kn, fn in (("fwd", fwd), ("dX", dx), ("dW", dw), ("fwd6", fwd_s), ("dX6", dx_s), ("dW6", dw_s)) + h2:
            ms = timeit(fn, args.reps)
            print(f"{name:13s} {kn:4s} N={n} Fi={fi}x{nseg} Fo={fo}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s")

    rest, _, rig = synth.make_batch(32)
    for name, b in (("soft", rest), ("rigid", rig)):
        n, e = b.x.shape[0], b.edge_index.shape[1]
        gi = GraphIndex(b.edge_index.to(dev), n)
        for f in (256, 21, 25, 84, 100):
            slab = torch.randn(n, 4 * f, device=dev)
            x, y = slab[:, :f], slab[:, f:2 * f]
            ms = timeit(lambda: ops.hop(gi.fwd, x, out=y), args.reps)
            nbytes = e * (8 + 4 * f) + n * (4 * f + 4)
            print(f"hop {name:6s} F={f:4d} (ld {4 * f}): {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:9.1f} GB/s algorithmic")


if __name__ == "__main__":
    main()
