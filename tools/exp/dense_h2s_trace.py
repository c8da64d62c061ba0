#!/usr/bin/env python3
"""Per-stage timeline of tools/exp/dense_h2s.hip (-DH2S_TRACE=1): when the MFMA waves and the data-moving waves reach each
stage's barrier (100 MHz wall clock, wave 0 / wave 4 of every workgroup), soft layer-2 block on a cold slab."""
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


def main():
    so = os.path.join(HERE, "libdenseh2s_trace.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-DH2S_TRACE=" + os.environ.get("H2S_TRACE", "1"), "-I" + os.path.join(ROOT, "include")] + EXTRA + [os.path.join(HERE, SRC), "-o", so])
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
        tr = torch.zeros(nwg, 2, 160, dtype=torch.int64, device=dev)
        X.h2s_set_trace(tr.data_ptr())
        for rep in range(6):                      # the last repetition is the one read (slabs rotate: cold operands)
            sl, rm = slabs[rep % 3], rowmax[rep % 3]
            X.h2s_run(sl.data_ptr(), sl.stride(0), wimg.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), fo, n, k, fo,
                      rm.data_ptr(), wmax.data_ptr(), st)
        torch.cuda.synchronize()
        X.h2s_set_trace(None)
        t = tr.cpu().double() / 100.0              # us
        t0 = float(t[:, :, 0].min())
        c, p = t[:, 0] - t0, t[:, 1] - t0
        end = float(c[:, 71].max())
        print(f"N={n}: {nwg} workgroups; first wave start -> last store drained {end:.1f} us")
        print(f"  workgroup start (consumer wave):  median {float(c[:, 0].median()):.2f}  max {float(c[:, 0].max()):.2f} us")
        print(f"  prologue (start -> barrier P released): median {float((c[:, 2] - c[:, 0]).median()):.2f} us "
              f"(producer reaches P after {float((p[:, 1] - p[:, 0]).median()):.2f})")
        loop = c[:, 5 + 2 * (nst - 1)] - c[:, 2]
        print(f"  loop: median {float(loop.median()):.2f}  max {float(loop.max()):.2f} us")
        epi = c[:, 70] - c[:, 5 + 2 * (nst - 1)]
        drain = c[:, 71] - c[:, 70]
        print(f"  epilogue issue: median {float(epi.median()):.2f} us; drain {float(drain.median()):.2f} us; "
              f"workgroup end median {float(c[:, 71].median()):.2f} max {end:.2f}")
        # per iteration: period, consumer busy (release -> arrive), producer busy
        rel_prev = c[:, 2]
        rows = []
        for it in range(nst):
            ca, pa, rel = c[:, 4 + 2 * it], p[:, 4 + 2 * it], c[:, 5 + 2 * it]
            prel_prev = p[:, 2] if it == 0 else p[:, 5 + 2 * (it - 1)]
            rows.append((float((rel - rel_prev).median()), float((ca - rel_prev).median()), float((pa - prel_prev).median()),
                         float(((pa > ca).double()).mean())))
            rel_prev = rel
        if os.environ.get("H2S_TRACE") == "2":
            print("  producer, even iterations: release -> loads issued -> loads landed -> LDS stores done -> barrier")
            for it in range(0, nst - 4, 2):
                prel = p[:, 2] if it == 0 else p[:, 5 + 2 * (it - 1)]
                a, b, d, e = p[:, 80 + 2 * it], p[:, 81 + 2 * it], p[:, 82 + 2 * it], p[:, 4 + 2 * it]
                print(f"  {it:2d}: {float((a - prel).median()):5.2f} {float((b - a).median()):5.2f} "
                      f"{float((d - b).median()):5.2f} {float((e - d).median()):5.2f}")
        print("  it: period  consumer-busy  producer-busy  frac(producer last)")
        for it, r in enumerate(rows):
            print(f"  {it:2d}: {r[0]:6.2f} {r[1]:6.2f} {r[2]:6.2f} {r[3]:5.2f}")


if __name__ == "__main__":
    main()
