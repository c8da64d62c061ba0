#!/usr/bin/env python3
"""Attention forward of one head at the B = 32 shape (32,768 soft x 24,384 rigid nodes, d = 256): the flash-style
launch (dc_attn_flash_fwd) against the blocked three-launch form, ms per forward (operands prepared outside)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import attention  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ns, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 24384)
    torch.manual_seed(0)
    q = torch.randn(ns, 256, device=dev) * 0.3
    k = torch.randn(nr, 256, device=dev) * 0.3
    v = torch.randn(nr, 256, device=dev)
    res = {}
    for flash in (True, False):
        attention.FLASH = flash
        with torch.no_grad():
            for _ in range(2):
                o = attention.attention_core(q, k, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 5
            for _ in range(n):
                o = attention.attention_core(q, k, v)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[flash] = o
        flop = 2 * 2 * ns * nr * 256 * 3
        print(f"{'flash  ' if flash else 'blocked'} forward: {ms:8.3f} ms   ({flop / ms / 1e9:7.1f} TFLOP/s of fp16 products, incl. operand prep)")
    # the flash launch alone, operands prepared once
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    L = _lib.lib()
    st = current_stream_ptr(dev)
    nrp = attention._ceil_keys(nr)
    kp, vp = attention._pad_rows(k - k.mean(dim=0, keepdim=True), nrp), attention._pad_rows(v, nrp)
    kmax, kimg, _, _ = attention._prep(L, kp, False, st)
    _, _, vtimg, vtmax = attention._prep(L, vp, True, st)
    kuns = torch.empty_like(kmax)
    _lib.check(L.dc_attn_flash_prep(vtimg.data_ptr(), 256, nrp, kmax.data_ptr(), kuns.data_ptr(), st), "dc_attn_flash_prep")
    qmax = attention._rowabsmax(L, q, st)
    o = torch.empty((ns, 256), device=dev)
    lse = torch.empty(ns, device=dev)

    def launch():
        _lib.check(L.dc_attn_flash_fwd(q.data_ptr(), 256, qmax.data_ptr(), kimg.data_ptr(), kuns.data_ptr(),
                                       vtimg.data_ptr(), vtmax.data_ptr(), ns, nr, nrp, 256, o.data_ptr(), 256,
                                       lse.data_ptr(), st), "dc_attn_flash_fwd")
    launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flop = 2 * 2 * ns * nr * 256 * 3
    print(f"dc_attn_flash_fwd alone: {ms:8.3f} ms   ({flop / ms / 1e9:7.1f} TFLOP/s of fp16 products = {flop / ms / 1e9 / 2500:.3f} of 2.5 PF)")
    d = (res[True] - res[False]).abs().max().item() / res[False].abs().max().item()
    print(f"max |flash - blocked| / max |blocked| = {d:.2e}")


if __name__ == "__main__":
    main()
