#!/usr/bin/env python3
"""Time the wide forward-shaped fp16x2 block (dc_tag_linear_fwd_h2p -> k_fwd_h2d) and the wide dW block at the B = 32 shapes,
on rotating slabs (operands from beyond the Infinity Cache), HIP events.  For A/B runs of variant libraries:
    python tools/exp/run_with_lib.py <lib.so> tools/r06/dense_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _i64_array, _ptr_array  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = current_stream_ptr(dev)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    nrot = 4
    for name, n in (("soft", 32768), ("rigid", 24384)):
        fi = fo = 256
        nseg = 4
        slabs = [ops._alloc_slab(n, nseg * fi, dev).normal_() for _ in range(nrot)]
        ws = [torch.randn(fo, fi, device=dev) / 16 for _ in range(nseg)]
        bias = torch.randn(fo, device=dev)
        outs = [torch.empty(n, fo, device=dev) for _ in range(nrot)]
        rowmax = [s.abs().amax(1).contiguous() for s in slabs]
        wmax = ops.weight_rowmax(ws)
        wimg, wtimg = torch.empty(fo, nseg * fi, device=dev), torch.empty(fi, nseg * fo, device=dev)
        wtmax = torch.empty(fi, device=dev)
        L.dc_tag_weight_prep(_ptr_array(ws), nseg, fo, fi, wmax.data_ptr(), wimg.data_ptr(), wtimg.data_ptr(), wtmax.data_ptr(), st)
        gws = [torch.empty(fo, fi, device=dev) for _ in range(nseg)]
        gb = torch.empty(fo, device=dev)
        nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        gs = [ops._alloc_slab(n, nseg * fo, dev).normal_() for _ in range(nrot)]
        gmax = [g[:, :fo].abs().amax(1).contiguous() for g in gs]

        def fwd(i):
            s = slabs[i % nrot]
            _lib.check(L.dc_tag_linear_fwd_h2p(s.data_ptr(), s.stride(0), wimg.data_ptr(), bias.data_ptr(), 1,
                                               outs[i % nrot].data_ptr(), fo, n, nseg * fi, fo, rowmax[i % nrot].data_ptr(),
                                               wmax.data_ptr(), None, 0, st), "fwd")

        def dw(i):
            s, g = slabs[i % nrot], gs[i % nrot]
            xs = [s[:, j * fi:(j + 1) * fi] for j in range(nseg)]
            _lib.check(L.dc_tag_linear_bwd_dw_h2(g.data_ptr(), g.stride(0), None, fo, _ptr_array(xs),
                                                 _i64_array([s.stride(0)] * nseg), nseg, _ptr_array(gws), nseg, fi,
                                                 gb.data_ptr(), 0, scratch.data_ptr(), nb, n, fi, fo,
                                                 gmax[i % nrot].data_ptr(), rowmax[i % nrot].data_ptr(), st), "dw")
        for tag, fn in (("fwd", fwd), ("dW+reduce", dw)):
            for i in range(4):
                fn(i)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e3)
            ts.sort()
            fl = 3 * 2.0 * n * fi * nseg * fo
            print(f"{name:5s} {tag:10s} {ts[2]:7.1f} us (min {ts[0]:.1f})  {fl / ts[2] / 1e6:7.1f} TF/s executed = {fl / ts[2] / 1e6 / 2500:.3f} of 2.5 PF",
                  flush=True)


if __name__ == "__main__":
    main()
