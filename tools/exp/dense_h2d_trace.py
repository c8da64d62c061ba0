#!/usr/bin/env python3
"""Where a launch of tools/exp/dense_h2d.hip spends its time outside the main loop (-DH2S_TRACE=1): six launches back to
back from one hipGraph on cold rotating slabs, 100 MHz wall clock per workgroup (start, prologue barrier, loop end,
epilogue issued, stores drained) - spans, gaps between launches, prologue / epilogue lengths."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import _lib, ops  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402
from deformcontact_amd.ops import _ptr_array  # noqa: E402

EXTRA = os.environ.get("H2S_FLAGS", "").split()
SRC = os.environ.get("H2S_SRC", "dense_h2d.hip")


def main():
    so = os.path.join(HERE, "libdenseh2d_trace.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-DH2S_TRACE=1", "-I" + os.path.join(ROOT, "include")] + EXTRA + [os.path.join(HERE, SRC), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.h2s_run.argtypes = [vp, i64, vp, vp, ci, vp, i64, i64, i64, i64, vp, vp, vp]
    X.h2s_set_trace.argtypes = [vp]
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n in (32768, 24384):
        k, fo = 1024, 256
        slabs = [ops._alloc_slab(n, k, dev).normal_() for _ in range(3)]
        ws = [torch.randn(fo, 256, device=dev) / 16 for _ in range(4)]
        bias = torch.randn(fo, device=dev)
        rowmax = [s.abs().amax(1).contiguous() for s in slabs]
        wmax = ops.weight_rowmax(ws)
        wimg = torch.empty(fo, k, device=dev)
        st = current_stream_ptr(dev)
        L.dc_tag_weight_prep(_ptr_array(ws), 4, fo, 256, wmax.data_ptr(), wimg.data_ptr(), None, None, st)
        out = torch.empty(n, fo, device=dev)
        nwg = (n + 127) // 128
        nl = 6
        tr = torch.zeros(nl, nwg, 8, dtype=torch.int64, device=dev)
        X.h2s_set_trace(tr.data_ptr())

        def run():
            s = current_stream_ptr(dev)
            for rep in range(nl):
                sl, rm = slabs[rep % 3], rowmax[rep % 3]
                X.h2s_run(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1 | (rep << 8), out.data_ptr(), fo, n, k,
                          fo, rm.data_ptr(), wmax.data_ptr(), s)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        X.h2s_set_trace(None)
        t = tr.cpu().double() / 100.0              # us
        t = t - float(t[0, :, 0].min())
        print(f"N={n}: {nwg} workgroups, {nl} launches from one hipGraph")
        prev_end = None
        for i in range(nl):
            a = t[i]
            start, end = float(a[:, 0].min()), float(a[:, 4].max())
            gap = "" if prev_end is None else f" gap {start - prev_end:5.2f}"
            print(f"  launch {i}: span {end - start:6.2f}{gap} | starts spread {float(a[:, 0].max() - a[:, 0].min()):.2f} | "
                  f"start->P {float((a[:, 1] - a[:, 0]).median()):.2f} (weights there after "
                  f"{float((a[:, 5] - a[:, 0]).median()):.2f}, x after {float((a[:, 6] - a[:, 0]).median()):.2f}) | "
                  f"loop {float((a[:, 2] - a[:, 1]).median()):.2f} (max {float((a[:, 2] - a[:, 1]).max()):.2f}) | "
                  f"epilogue {float((a[:, 3] - a[:, 2]).median()):.2f} + drain {float((a[:, 4] - a[:, 3]).median()):.2f} | "
                  f"ends spread {float(a[:, 4].max() - a[:, 4].min()):.2f}")
            prev_end = end


if __name__ == "__main__":
    main()
