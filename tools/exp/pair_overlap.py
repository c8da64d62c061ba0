#!/usr/bin/env python3
"""Which kernels of the two branches gain from running side by side?  Pairs of layer-2 launches of the soft and the
rigid branch on two streams inside one hipGraph (fork / join), against the same two run one after the other.
D = forward-shaped dense block (k_fwd_h2w), W = dW block (+ slab reduce), H = chain of three F=256 hops."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _i64_array, _ptr_array  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    rest, _, rig = synth.make_batch(32)
    side = torch.cuda.Stream()
    br = {}
    for name, b in (("s", rest), ("r", rig)):
        n = b.x.shape[0]
        g = GraphIndex(b.edge_index.to(dev), n)
        slab, gslab = ops._alloc_slab(n, 1024, dev).normal_(), ops._alloc_slab(n, 1024, dev).normal_()
        hslab = ops._alloc_slab(n, 1024, dev).normal_()
        ws = [torch.randn(256, 256, device=dev) / 16 for _ in range(4)]
        bias, out = torch.randn(256, device=dev), torch.empty(n, 256, device=dev)
        rowmax, wmax = slab.abs().amax(1).contiguous(), ops.weight_rowmax(ws)
        wimg = torch.empty(256, 1024, device=dev)
        L.dc_tag_weight_prep(_ptr_array(ws), 4, 256, 256, wmax.data_ptr(), wimg.data_ptr(), None, None,
                             current_stream_ptr(dev))
        gws = [torch.empty(256, 256, device=dev) for _ in range(4)]
        gb = torch.empty(256, device=dev)
        nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, 256, 256, 4)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        gmax = gslab.abs().amax(1).contiguous()
        hmax = torch.zeros(n, device=dev)
        pa_x = _ptr_array([slab[:, s * 256:(s + 1) * 256] for s in range(4)])
        pa_ld, pa_gw = _i64_array([slab.stride(0)] * 4), _ptr_array(gws)
        keep = (ws, bias, gws, gb, scratch, pa_x, pa_ld, pa_gw, wimg, rowmax, wmax, gmax, hmax)

        def D(n=n, slab=slab, wimg=wimg, bias=bias, out=out, rowmax=rowmax, wmax=wmax):
            L.dc_tag_linear_fwd_h2p(slab.data_ptr(), slab.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(),
                                    256, n, 1024, 256, rowmax.data_ptr(), wmax.data_ptr(), None, 0, current_stream_ptr(dev))

        def W(n=n, gslab=gslab, pa_x=pa_x, pa_ld=pa_ld, pa_gw=pa_gw, gb=gb, scratch=scratch, nb=nb, gmax=gmax,
              rowmax=rowmax):
            L.dc_tag_linear_bwd_dw_h2(gslab.data_ptr(), gslab.stride(0), None, 256, pa_x, pa_ld, 4, pa_gw, 4, 256,
                                      gb.data_ptr(), 0, scratch.data_ptr(), nb, n, 256, 256, gmax.data_ptr(),
                                      rowmax.data_ptr(), current_stream_ptr(dev))

        def H(g=g, hslab=hslab, hmax=hmax):
            ops.chained_hops(g, hslab, 256, 3, backward=False, rowmax=hmax)
        br[name] = {"D": D, "W": W, "H": H, "keep": keep}

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(4):
                fn()
        gr.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        return ts[2]

    def both(a, b):
        def fn():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b()
            a()
            main.wait_stream(side)
        return fn

    for ka, kb in (("D", "H"), ("H", "D"), ("W", "H"), ("H", "W"), ("D", "D"), ("H", "H"), ("D", "W")):
        a, b = br["s"][ka], br["r"][kb]
        ta, tb = timed(a), timed(b)
        tboth = timed(both(a, b))
        print(f"soft {ka} {ta:6.1f} us  rigid {kb} {tb:6.1f} us  one after the other {ta + tb:6.1f}  side by side {tboth:6.1f} us "
              f"({tboth / (ta + tb):.2f} of the sum; the longer one alone {max(ta, tb):.1f})")


if __name__ == "__main__":
    main()
