#!/usr/bin/env python3
"""Drive tools/exp/dense_h2s.hip (128 x 256 tiles, MFMA waves + data-moving waves) against the product's forward-shaped
block: bit-identity, then time on cold rotating slabs (soft N=32768 / rigid N=24384, K=1024, Fo=256), the product's
kernel timed the same way beside it.  Arguments: H2S_ABL values to build and time (timing-only builds; default 0)."""
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

SRC = os.environ.get("H2S_SRC", "dense_h2s.hip")
EXTRA = os.environ.get("H2S_FLAGS", "").split()


def build(abl):
    so = os.path.join(HERE, f"libdenseh2s_{abl}.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           f"-DH2S_ABL={abl}", "-I" + os.path.join(ROOT, "include")] + EXTRA +
                          [os.path.join(HERE, SRC), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.h2s_run.argtypes = [vp, i64, vp, vp, ci, vp, i64, i64, i64, i64, vp, vp, vp]
    X.h2s_set_dbg.argtypes = [vp]
    return X


def timed(run):
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
    return ts[3] * 1e3


def main():
    abls = [int(a) for a in sys.argv[1:]] or [0]
    libs = {a: build(a) for a in abls}
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n in (32768, 24384, 24000):
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

        def prod():
            s = current_stream_ptr(dev)
            for sl, rm in zip(slabs, rowmax):
                L.dc_tag_linear_fwd_h2p(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo,
                                        n, k, fo, rm.data_ptr(), wmax.data_ptr(), None, 0, s)
        line = f"N={n}: product {timed(prod):6.1f} us |"
        for a, X in libs.items():
            if a == 0:
                out.zero_()
                rc = X.h2s_run(slabs[0].data_ptr(), slabs[0].stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo,
                               n, k, fo, rowmax[0].data_ptr(), wmax.data_ptr(), st)
                torch.cuda.synchronize()
                same = rc == 0 and torch.equal(ref, out)
                line += " bit-identical" if same else f" DIFFERENT (max {float((ref - out).abs().max()):.3e})"

            def run():
                s = current_stream_ptr(dev)
                for sl, rm in zip(slabs, rowmax):
                    X.h2s_run(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n, k, fo,
                              rm.data_ptr(), wmax.data_ptr(), s)
            us = timed(run)
            dbg = torch.zeros(2 * ((n + 127) // 128), dtype=torch.int64, device=dev)
            X.h2s_set_dbg(dbg.data_ptr())
            for _ in range(4):
                run()
            torch.cuda.synchronize()
            X.h2s_set_dbg(None)
            d = dbg.view(-1, 2).double()
            ghz = float((d[:, 0] / d[:, 1].clamp_min(1)).median()) * 0.1
            loop_us = float(d[:, 1].median()) / 100.0
            line += f" abl{a}: {us:6.1f} us ({2.0 * n * k * fo * 3 / us / 1e6 / 2500:.3f}; loop {loop_us:.1f} us @ {ghz:.2f} GHz)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
