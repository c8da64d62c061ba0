#!/usr/bin/env python3
"""Layer-1 dense blocks of the encoder (one K segment over the concatenated slab, K = 96 / 112, Fo = 256): the six-product
bf16 split the product uses against the three-product fp16x2 entries, us per launch (graph-replayed, rotating buffers)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import _lib  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _i64_array, _ptr_array  # noqa: E402


def timeit(fn, reps=20):
    dev = torch.device("cuda:0")
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n, wpad in ((32768, 96), (24384, 112), (24384, 128)):
        fo = 256
        nrot = 6
        slabs = [torch.randn(n, wpad, device=dev) for _ in range(nrot)]
        outs = [torch.empty(n, fo, device=dev) for _ in range(nrot)]
        gs = [torch.randn(n, fo, device=dev) for _ in range(nrot)]
        w = torch.randn(fo, wpad, device=dev) * 0.1
        bias = torch.zeros(fo, device=dev)
        xmax = [s.abs().amax(dim=1).contiguous() for s in slabs]
        gmax = [s.abs().amax(dim=1).contiguous() for s in gs]
        wmax = w.abs().amax(dim=1).contiguous()
        gw = torch.empty(fo, wpad, device=dev)
        nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, wpad, fo, 1)
        scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        it = [0]

        def fwd(which):
            i = it[0] = (it[0] + 1) % nrot
            st = current_stream_ptr(dev)
            a = (_ptr_array([slabs[i]]), _i64_array([wpad]), _ptr_array([w]), 1, bias.data_ptr(), 1, outs[i].data_ptr(),
                 fo, n, wpad, fo)
            if which == "split6":
                _lib.check(L.dc_tag_linear_fwd_split(*a, 6, st), "fwd_split")
            else:
                _lib.check(L.dc_tag_linear_fwd_h2(*a, xmax[i].data_ptr(), wmax.data_ptr(), st), "fwd_h2")

        def dw(which):
            i = it[0] = (it[0] + 1) % nrot
            st = current_stream_ptr(dev)
            a = (gs[i].data_ptr(), fo, outs[i].data_ptr(), fo, _ptr_array([slabs[i]]), _i64_array([wpad]), 1,
                 _ptr_array([gw]), 1, wpad, None, 0, scratch.data_ptr(), nb, n, wpad, fo)
            if which == "split6":
                _lib.check(L.dc_tag_linear_bwd_dw_split(*a, 6, st), "dw_split")
            else:
                _lib.check(L.dc_tag_linear_bwd_dw_h2(*a, gmax[i].data_ptr(), xmax[i].data_ptr(), st), "dw_h2")

        for name, f in (("forward", fwd), ("dW     ", dw)):
            r = {k: timeit(lambda: f(k)) for k in ("split6", "h2")}
            print(f"N={n} K={wpad:3d} Fo={fo} {name}: six-product split {r['split6']:6.2f} us, fp16x2 {r['h2']:6.2f} us")


if __name__ == "__main__":
    main()
