#!/usr/bin/env python3
"""Drive tools/exp/dense_presplit.hip: the wide forward block with BOTH operands pre-split (pure LDS-DMA loop) against the
product's k_fwd_h2w - bit-identity, then time on rotating slabs (soft N = 32768 / rigid N = 24384, K = 1024, Fo = 256)."""
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
    so = os.path.join(HERE, "libdensepresplit.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "dense_presplit.hip"), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.presplit_run.argtypes = [vp, vp, vp, vp, vp, ci, vp, i64, i64, i64, i64, vp]
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n in (32768, 24384):
        k, fo = 1024, 256
        slabs = [torch.randn(n, k, device=dev) for _ in range(3)]                  # unpadded: weight_prep wants ld == K
        ws = [torch.randn(fo, 256, device=dev) / 16 for _ in range(4)]
        bias = torch.randn(fo, device=dev)
        st = current_stream_ptr(dev)
        wmax, wimg = torch.empty(fo, device=dev), torch.empty(fo, k, device=dev)
        L.dc_tag_weight_prep(_ptr_array(ws), 4, fo, 256, wmax.data_ptr(), wimg.data_ptr(), None, None, st)
        aimgs, amaxs = [], []
        for s in slabs:                                   # the activation image: the same record format, per-row scale
            am, ai = torch.empty(n, device=dev), torch.empty(n, k, device=dev)
            L.dc_tag_weight_prep(_ptr_array([s]), 1, n, k, am.data_ptr(), ai.data_ptr(), None, None, st)
            aimgs.append(ai), amaxs.append(am)
        ref, out = torch.empty(n, fo, device=dev), torch.empty(n, fo, device=dev)
        L.dc_tag_linear_fwd_h2p(slabs[0].data_ptr(), k, wimg.data_ptr(), bias.data_ptr(), 1, ref.data_ptr(), fo, n, k, fo,
                                amaxs[0].data_ptr(), wmax.data_ptr(), None, 0, st)
        rc = X.presplit_run(aimgs[0].data_ptr(), wimg.data_ptr(), amaxs[0].data_ptr(), wmax.data_ptr(), bias.data_ptr(), 1,
                            out.data_ptr(), fo, n, k, fo, st)
        torch.cuda.synchronize()
        assert rc == 0
        same = torch.equal(ref, out)
        err = float((ref - out).abs().max() / ref.abs().max())

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
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

        def run_pre():
            s = current_stream_ptr(dev)
            for ai, am in zip(aimgs, amaxs):
                X.presplit_run(ai.data_ptr(), wimg.data_ptr(), am.data_ptr(), wmax.data_ptr(), bias.data_ptr(), 1,
                               out.data_ptr(), fo, n, k, fo, s)

        def run_prod():
            s = current_stream_ptr(dev)
            for sl, am in zip(slabs, amaxs):
                L.dc_tag_linear_fwd_h2p(sl.data_ptr(), k, wimg.data_ptr(), bias.data_ptr(), 1, ref.data_ptr(), fo, n, k, fo,
                                        am.data_ptr(), wmax.data_ptr(), None, 0, s)
        t_pre, t_prod = timed(run_pre), timed(run_prod)
        flop = 3 * 2.0 * n * k * fo
        print(f"N={n}: bit-identical {same} (max rel diff {err:.2e}); pre-split / DMA-only loop {t_pre:6.1f} us "
              f"({flop / t_pre / 1e6 / 2500:.3f} of 2.5 PF), product k_fwd_h2w {t_prod:6.1f} us ({flop / t_prod / 1e6 / 2500:.3f})")


if __name__ == "__main__":
    main()
