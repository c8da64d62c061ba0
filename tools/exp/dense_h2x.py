#!/usr/bin/env python3
"""Drive tools/exp/dense_h2x.hip (256 x 256 tiles, one wave per SIMD) against the product's forward-shaped block:
bit-identity, then time on cold rotating slabs (soft N=32768 / rigid N=24384, K=1024, Fo=256)."""
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


def main():
    so = os.path.join(HERE, "libdenseh2x.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "dense_h2x.hip"), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.h2x_run.argtypes = [vp, i64, vp, vp, ci, vp, i64, i64, i64, i64, vp, vp, vp]
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n in (32768, 24384, 57344):              # 57344 = both branches as ONE problem (224 tiles of 256 rows)
        k, fo = 1024, 256
        slabs = [ops._alloc_slab(n, k, dev).normal_() for _ in range(3)]
        ws = [torch.randn(fo, 256, device=dev) / 16 for _ in range(4)]
        bias = torch.randn(fo, device=dev)
        rowmax = [s.abs().amax(1).contiguous() for s in slabs]
        wmax = ops.weight_rowmax(ws)
        wimg = torch.empty(fo, k, device=dev)
        st = current_stream_ptr(dev)
        L.dc_tag_weight_prep(_ptr_array(ws), 4, fo, 256, wmax.data_ptr(), wimg.data_ptr(), None, None, st)
        ref, out = torch.empty(n, fo, device=dev), torch.empty(n, fo, device=dev)
        L.dc_tag_linear_fwd_h2p(slabs[0].data_ptr(), slabs[0].stride(0), wimg.data_ptr(), bias.data_ptr(), 1, ref.data_ptr(),
                                fo, n, k, fo, rowmax[0].data_ptr(), wmax.data_ptr(), None, 0, st)
        rc = X.h2x_run(slabs[0].data_ptr(), slabs[0].stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n, k,
                       fo, rowmax[0].data_ptr(), wmax.data_ptr(), st)
        torch.cuda.synchronize()
        assert rc == 0 and torch.equal(ref, out), "256 x 256 tiles: not bit-identical to the product kernel"

        def run():
            s = current_stream_ptr(dev)
            for sl, rm in zip(slabs, rowmax):
                X.h2x_run(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n, k, fo,
                          rm.data_ptr(), wmax.data_ptr(), s)
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

        def run_prod():
            s = current_stream_ptr(dev)
            for sl, rm in zip(slabs, rowmax):
                L.dc_tag_linear_fwd_h2p(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, ref.data_ptr(), fo, n, k,
                                        fo, rm.data_ptr(), wmax.data_ptr(), None, 0, s)
        run_prod()
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            run_prod()
        tp = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g2.replay()
            e1.record()
            torch.cuda.synchronize()
            tp.append(e0.elapsed_time(e1) / 30)
        tp.sort()
        print(f"N={n}: product kernel (k_fwd_h2d, 128 x 256 tiles) {tp[3] * 1e3:6.1f} us", flush=True)
        tiles = (n + 255) // 256
        print(f"N={n}: bit-identical; {us:6.1f} us on {tiles} workgroups = "
              f"{2.0 * 256 * k * fo * 3 / us / 1e6:5.2f} TF/s per CU ({2.0 * 256 * k * fo * 3 / us / 1e6 / (2500 / 256):.2f} of a CU's peak)")


if __name__ == "__main__":
    main()
