#!/usr/bin/env python3
"""Per-stage timeline of tools/exp/dense_h2r.hip (-DH2S_TRACE=1|2): when the MFMA waves (role 0), the weight waves (1) and
the x waves (2) reach each stage's barrier (100 MHz wall clock, first wave of each role), soft / rigid layer-2 block."""
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
TR = os.environ.get("H2S_TRACE", "1")


def main():
    so = os.path.join(HERE, "libdenseh2r_trace.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-DH2S_TRACE=" + TR, "-I" + os.path.join(ROOT, "include")] + EXTRA +
                          [os.path.join(HERE, "dense_h2r.hip"), "-o", so])
    X = ctypes.CDLL(so)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    X.h2s_run.argtypes = [vp, i64, vp, vp, ci, vp, i64, i64, i64, i64, vp, vp, vp]
    X.h2s_set_trace.argtypes = [vp]
    dev = torch.device("cuda:0")
    L = _lib.lib()
    for n in (32768, 24384):
        k, fo = 1024, 256
        nst = k // 32
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
        tr = torch.zeros(nwg, 3, 160, dtype=torch.int64, device=dev)
        X.h2s_set_trace(tr.data_ptr())
        for rep in range(6):
            sl, rm = slabs[rep % 3], rowmax[rep % 3]
            X.h2s_run(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n, k, fo,
                      rm.data_ptr(), wmax.data_ptr(), st)
        torch.cuda.synchronize()
        X.h2s_set_trace(None)
        t = tr.cpu().double() / 100.0              # us
        t0 = float(t[:, :, 0].min())
        c, w, x = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
        end = float(c[:, 71].max())
        print(f"N={n}: {nwg} workgroups; first wave start -> last store drained {end:.1f} us")
        print(f"  prologue (start -> barrier P released): median {float((c[:, 2] - c[:, 0]).median()):.2f} us "
              f"(weight waves reach P after {float((w[:, 1] - w[:, 0]).median()):.2f}, x waves after {float((x[:, 1] - x[:, 0]).median()):.2f})")
        last = 5 + 2 * (nst - 1)
        loop = c[:, last] - c[:, 2]
        print(f"  loop: median {float(loop.median()):.2f}  max {float(loop.max()):.2f} us")
        print(f"  epilogue issue: median {float((c[:, 70] - c[:, last]).median()):.2f} us; drain "
              f"{float((c[:, 71] - c[:, 70]).median()):.2f} us; workgroup end median {float(c[:, 71].median()):.2f} max {end:.2f}")
        print("  it: period | busy (release -> at barrier): MFMA  weights  x | who is last (MFMA / weights / x)")
        for it in range(nst):
            def rel(a):
                return a[:, 2] if it == 0 else a[:, 5 + 2 * (it - 1)]
            ca, wa, xa = c[:, 4 + 2 * it], w[:, 4 + 2 * it], x[:, 4 + 2 * it]
            lastw = torch.stack([ca, wa, xa], 1).argmax(1)
            fr = [float((lastw == i).double().mean()) for i in range(3)]
            extra = ""
            if TR == "2" and it + 8 <= nst:
                a, b = x[:, 80 + 2 * it], x[:, 81 + 2 * it]
                extra = (f" | x: issue {float((a - rel(x)).median()):.2f} wait {float((b - a).median()):.2f} "
                         f"split+store {float((xa - b).median()):.2f}")
            print(f"  {it:2d}: {float((c[:, 5 + 2 * it] - rel(c)).median()):5.2f} | {float((ca - rel(c)).median()):5.2f} "
                  f"{float((wa - rel(w)).median()):5.2f} {float((xa - rel(x)).median()):5.2f} | "
                  f"{fr[0]:.2f} {fr[1]:.2f} {fr[2]:.2f}{extra}")


if __name__ == "__main__":
    main()
