#!/usr/bin/env python3
"""Drive tools/exp/astream.hip: rate at which 128-row tiles can stream a cold [32768, 1024] fp32 slab."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from deformcontact_amd import ops  # noqa: E402
from deformcontact_amd.graph import current_stream_ptr  # noqa: E402


def main():
    so = os.path.join(HERE, "libastream.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
                           os.path.join(HERE, "astream.hip"), "-o", so])
    L = ctypes.CDLL(so)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.astream_run.argtypes = [ctypes.c_int, ctypes.c_int, vp, i64, i64, ctypes.c_int, vp, vp]
    dev = torch.device("cuda:0")
    n, k = 32768, 1024
    slabs = [ops._alloc_slab(n, k, dev).normal_() for _ in range(3)]
    out = torch.empty(n // 128 * 512, device=dev)
    for ch, d in ((128, 2), (128, 3), (128, 4), (128, 6), (128, 8), (256, 1), (256, 2), (256, 3), (256, 4),
                  (512, 1), (512, 2), (512, 3), (1024, 1), (1024, 2)):
        def run():
            for sl in slabs:
                rc = L.astream_run(ch, d, sl.data_ptr(), sl.stride(0), n, k, out.data_ptr(), current_stream_ptr(dev))
                assert rc == 0
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30)
        ts.sort()
        us = ts[2] * 1e3
        print(f"chunk {ch:5d} B/row  depth {d}  ({ch * 128 * d // 1024:4d} KB in flight per CU): {us:6.1f} us  "
              f"{n * k * 4 / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
