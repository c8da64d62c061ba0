#!/usr/bin/env python3
"""Time the forward-shaped fp16x2 dense block (soft layer 2: N=32768, K=4x256, Fo=256) from a hipGraph.

    DC_H2_ABL=<n> python tools/exp/dense_abl.py      (ablation builds only; the product ignores it)
    python tools/exp/dense_abl.py --wide              (time dc_tag_linear_fwd_h2p under DC_H2_WIDE=0/1 is set by env)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _ptr_array  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    res = []
    for n in (32768, 24384):
        fi, fo, nseg = 256, 256, 4
        slabs = [ops._alloc_slab(n, nseg * fi, dev).normal_() for _ in range(3)]   # rotate: cold operands
        ws = [torch.randn(fo, fi, device=dev) / 16 for _ in range(nseg)]
        bias, out = torch.randn(fo, device=dev), torch.empty(n, fo, device=dev)
        rowmax = [s.abs().amax(1).contiguous() for s in slabs]
        wmax = ops.weight_rowmax(ws)
        wimg = torch.empty(fo, nseg * fi, device=dev)
        st = current_stream_ptr(dev)
        L.dc_tag_weight_prep(_ptr_array(ws), nseg, fo, fi, wmax.data_ptr(), wimg.data_ptr(), None, None, st)
        torch.cuda.synchronize()

        def run():
            s = current_stream_ptr(dev)
            for sl, rm in zip(slabs, rowmax):
                L.dc_tag_linear_fwd_h2p(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1,
                                        out.data_ptr(), fo, n, nseg * fi, fo, rm.data_ptr(), wmax.data_ptr(), None, 0, s)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30)
        ts.sort()
        us = ts[len(ts) // 2] * 1e3
        fl = 2.0 * n * nseg * fi * fo * 3
        res.append(f"N={n}: {us:6.1f} us  {fl / us / 1e6:6.0f} TF executed  frac {fl / us / 1e6 / 2500:.3f}")
        if "--check" in sys.argv:
            ref = torch.relu(slabs[-1].double() @ torch.cat(ws, 1).double().t() + bias.double())
            err = ((out.double() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-30)).max().item()
            res[-1] += f"  max row-rel err {err:.2e}"
    print(f"ABL={os.environ.get('DC_H2_ABL', '0')} WIDE={os.environ.get('DC_H2_WIDE', '-')}: " + " | ".join(res))


if __name__ == "__main__":
    main()
