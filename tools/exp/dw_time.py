#!/usr/bin/env python3
"""Time the layer-2 dW block (fp16x2) on cold rotating operands from a hipGraph: DC_DW_WIDE=0/1."""
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
    res = []
    for n in (32768, 24384):
        fi = fo = 256
        nseg = 4
        slabs = [ops._alloc_slab(n, nseg * fi, dev).normal_() for _ in range(3)]
        gs = [torch.randn(n, fo, device=dev) for _ in range(3)]
        gws = [torch.empty(fo, fi, device=dev) for _ in range(nseg)]
        gb = torch.empty(fo, device=dev)
        nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        gmax = [g.abs().amax(1).contiguous() for g in gs]
        xmax = [s.abs().amax(1).contiguous() for s in slabs]
        pas = [_ptr_array([sl[:, s * fi:(s + 1) * fi] for s in range(nseg)]) for sl in slabs]
        pa_gw, pa_ld = _ptr_array(gws), _i64_array([slabs[0].stride(0)] * nseg)

        def run():
            st = current_stream_ptr(dev)
            for i in range(3):
                rc = L.dc_tag_linear_bwd_dw_h2(gs[i].data_ptr(), fo, None, fo, pas[i], pa_ld, nseg, pa_gw, nseg, fi,
                                               gb.data_ptr(), 0, scratch.data_ptr(), nb, n, fi, fo,
                                               gmax[i].data_ptr(), xmax[i].data_ptr(), st)
                assert rc == 0
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
        us = ts[3] * 1e3
        fl = 2.0 * n * nseg * fi * fo * 3
        res.append(f"N={n}: {us:6.1f} us (dW + reduce)  {fl / us / 1e6:6.0f} TF executed  frac {fl / us / 1e6 / 2500:.3f}")
    print(f"DC_DW_WIDE={os.environ.get('DC_DW_WIDE', '-')}: " + " | ".join(res))


if __name__ == "__main__":
    main()
