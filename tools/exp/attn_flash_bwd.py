#!/usr/bin/env python3
"""Attention backward of one head at the B = 32 shape: dc_attn_flash_ds alone (ms, PFLOP/s of fp16 products over its 4
GEMM-equivalents) and the whole backward (flash vs blocked)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import _lib, attention  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ns, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 24384)
    torch.manual_seed(0)
    q = torch.randn(ns, 256, device=dev) * 0.3
    k = torch.randn(nr, 256, device=dev) * 0.3
    v = torch.randn(nr, 256, device=dev)
    go = torch.randn(ns, 256, device=dev)
    L, st = _lib.lib(), current_stream_ptr(dev)
    nrp = attention._ceil_keys(nr)
    kp, vp = attention._pad_rows(k - k.mean(dim=0, keepdim=True), nrp), attention._pad_rows(v, nrp)
    kmax, kimg, _, _ = attention._prep(L, kp, False, st)
    vmax, vimg, _, _ = attention._prep(L, vp, False, st)
    kuns, vuns = torch.empty_like(kmax), torch.empty_like(vmax)
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, kmax.data_ptr(), kuns.data_ptr(), st), "prep")
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, vmax.data_ptr(), vuns.data_ptr(), st), "prep")
    qmax, gomax = attention._rowabsmax(L, q, st), attention._rowabsmax(L, go, st)
    lse = torch.logsumexp(q @ kp[:nr].t(), dim=1) if ns * nr <= (1 << 28) else torch.zeros(ns, device=dev) + 30.0
    p = torch.empty((ns, nrp), device=dev)
    ds = torch.empty((ns, nrp), device=dev)
    dsmax = torch.empty(ns, device=dev)

    def launch():
        _lib.check(L.dc_attn_flash_ds(q.data_ptr(), 256, qmax.data_ptr(), go.data_ptr(), 256, gomax.data_ptr(),
                                      kimg.data_ptr(), kuns.data_ptr(), vimg.data_ptr(), vuns.data_ptr(), lse.data_ptr(),
                                      ns, nr, nrp, 256, p.data_ptr(), ds.data_ptr(), nrp, dsmax.data_ptr(), None, None, st), "ds")
    launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flop = 4 * 2 * ns * nr * 256 * 3
    print(f"dc_attn_flash_ds alone: {ms:8.3f} ms   ({flop / ms / 1e12:6.3f} PFLOP/s of fp16 products = {flop / ms / 1e12 / 2.5:.3f} of 2.5 PF)")
    del p, ds
    for fb, single in ((True, True), (True, False), (False, False)):
        attention.FLASH_BWD, attention.FLASH_BWD_SINGLE = fb, single
        qq, kk, vv = (t.clone().requires_grad_() for t in (q, k, v))
        for it in range(3):
            o = attention.attention_core(qq, kk, vv)
            torch.cuda.synchronize()
            e0.record()
            o.backward(go)
            e1.record()
            torch.cuda.synchronize()
            qq.grad = kk.grad = vv.grad = None
        print(f"whole backward, {('flash, one sweep ' if single else 'flash, two sweeps') if fb else 'blocked          '}: {e0.elapsed_time(e1):8.3f} ms")


if __name__ == "__main__":
    main()
