#!/usr/bin/env python3
"""Layer-1 dense block (K = 96 / 128 after padding, Fo = 256) on the six-product bf16x3 kernels it uses today against the
three-product fp16x2 kernels of the wide layers (forced for K < 128): forward time from a hipGraph, and accuracy vs float64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
orig = ops._tag_uses_h2


def forced(fi, k, fo=None):
    if k == 0 and fo is not None and fi % 32 == 0 and fo % 16 == 0:
        return True
    return orig(fi, k, fo)


def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * reps) * 1e3


for n, k in ((32768, 96), (24384, 128), (24384, 96)):
    torch.manual_seed(0)
    x = torch.randn(n, k, device=dev) * torch.logspace(-2, 2, n, device=dev)[:, None]
    w = torch.randn(256, k, device=dev) / k ** 0.5
    b = torch.randn(256, device=dev)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    for name, fn in (("bf16x3 (6 products)", orig), ("fp16x2 (3 products)", forced)):
        ops._tag_uses_h2 = fn
        try:
            _lib.kernel_trace(True)
            with torch.no_grad():
                y = ops.dense_linear(x, w, b, relu=True)
            names = list(_lib.kernel_trace_counts())
            _lib.kernel_trace(False)
            err = float(((y.double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-30)).max())
            with torch.no_grad():
                t = timeit(lambda: ops.dense_linear(x, w, b, relu=True))
            print(f"N={n} K={k}: {name}: {t:.1f} us per call, per-row rel err vs float64 {err:.2e}, kernels {names}")
        except Exception as e:
            print(f"N={n} K={k}: {name}: {type(e).__name__}: {e}")
ops._tag_uses_h2 = orig
