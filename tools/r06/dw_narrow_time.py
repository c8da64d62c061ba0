#!/usr/bin/env python3
"""Time the first layers' weight-gradient block (dc_tag_linear_bwd_dw_split on the one-segment padded slab, mask applied while
staging, + its slab reduce) at the B = 32 shapes, rotating operands, HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _i64_array, _ptr_array  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
st = current_stream_ptr(dev)
for name, n, fi, wpad in (("soft", 32768, 21, 96), ("rigid", 24384, 25, 112)):
    fo, nrot = 256, 4
    slabs = [torch.randn(n, wpad, device=dev) for _ in range(nrot)]
    gs = [torch.randn(n, fo, device=dev) for _ in range(nrot)]
    outs = [torch.randn(n, fo, device=dev) for _ in range(nrot)]
    gws = [torch.empty(fo, fi, device=dev) for _ in range(4)]
    gb = torch.empty(fo, device=dev)
    nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, wpad, fo, 1)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)

    def dw(i):
        g, o, s = gs[i % nrot], outs[i % nrot], slabs[i % nrot]
        _lib.check(L.dc_tag_linear_bwd_dw_split(g.data_ptr(), fo, o.data_ptr(), fo, _ptr_array([s]), _i64_array([s.stride(0)]), 1,
                                                _ptr_array(gws), 4, fi, gb.data_ptr(), 0, scratch.data_ptr(), nb, n, wpad, fo, 6,
                                                st), "dw")
    for i in range(4):
        dw(i)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            dw(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"  {name}: dW + reduce {sorted(ts)[2]:.1f} us  (workspace {nb / 1e6:.1f} MB)", flush=True)
