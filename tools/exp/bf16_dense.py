#!/usr/bin/env python3
"""Time dc_tag_linear_fwd_bf16 on the configs[4] shape ([100k, 1024] x [256, 1024]^T) with padded / unpadded slab rows
and bf16 / fp32 output.  DC_BF16_X=0/1 picks the 128 x 128 or the 256 x 256 kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd import _lib  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L, st = _lib.lib(), current_stream_ptr(dev)
    n, k, fo = 100_000, 1024, 256
    w = (torch.randn(fo, k, device=dev) / 32).bfloat16()
    for pad in (0, 128):
        for out_bf16 in (1, 0):
            slabs = [torch.randn(n, k + pad, device=dev).bfloat16()[:, :k] for _ in range(3)]   # rotate: > Infinity Cache
            out = torch.empty(n, fo, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev)

            def run():
                for s in slabs:
                    _lib.check(L.dc_tag_linear_fwd_bf16(s.data_ptr(), s.stride(0), w.data_ptr(), None, 1, out.data_ptr(),
                                                        fo, out_bf16, n, k, fo, current_stream_ptr(dev)), "fwd")
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            byt = n * k * 2 + n * fo * (2 if out_bf16 else 4)
            print(f"lda {k + pad:5d} out {'bf16' if out_bf16 else 'fp32'}: {us:7.1f} us  {2.0 * n * k * fo / us / 1e6:7.1f} TF/s "
                  f"({2.0 * n * k * fo / us / 1e6 / 2500:.3f} of 2.5 PF)  {byt / us / 1e6:5.2f} TB/s of operand bytes")


if __name__ == "__main__":
    main()
