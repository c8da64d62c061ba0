#!/usr/bin/env python3
"""Error of the fp32-MFMA and the split-bf16x6 dense kernels against float64 (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deformcontact_amd import _lib
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
torch.manual_seed(0)
for name, mk in (("gauss", lambda *s: torch.randn(*s, device=dev)),
                 ("relu(gauss)*10", lambda *s: torch.relu(torch.randn(*s, device=dev)) * 10),
                 ("heavy-tail", lambda *s: torch.randn(*s, device=dev) * torch.exp(2 * torch.randn(*s, device=dev)))):
    n, fi, nseg, fo = 4096, 256, 4, 256
    slab = mk(n, nseg * fi)
    xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]; ld = [nseg * fi] * nseg
    ws = [torch.randn(fo, fi, device=dev) / 16 for _ in range(nseg)]
    bias = torch.randn(fo, device=dev)
    ref = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + bias.double()
    sc = ref.abs().max()
    for kind, np_ in (("fp32 mfma", 0), ("bf16x6   ", 6), ("bf16x3   ", 3), ("bf16x1   ", 1)):
        out = torch.empty(n, fo, device=dev)
        a = (_ptr_array(xs), _i64_array(ld), _ptr_array(ws), nseg, bias.data_ptr(), 0, out.data_ptr(), fo, n, fi, fo)
        L.dc_tag_linear_fwd_split(*a, np_, st) if np_ else L.dc_tag_linear_fwd(*a, st)
        err = (out.double() - ref).abs()
        print(f"{name:16s} {kind}: max|err|/max|ref| = {float(err.max() / sc):.3e}   rms rel = {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
    t = (slab.float() @ torch.cat(ws, 1).t() + bias)
    err = (t.double() - ref).abs()
    print(f"{name:16s} torch.mm  : max|err|/max|ref| = {float(err.max() / sc):.3e}   rms rel = {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
