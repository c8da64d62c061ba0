#!/usr/bin/env python3
"""Hop kernel beyond the Infinity Cache and on a heavy-tailed radius graph (BASELINE.json configs[4]).

    python tools/hop_stress.py

(1) 2048 everyday soft meshes batched: N = 2.1 M nodes, E = 12.6 M edges, x = 2.1 GB fp32 - the
    feature matrix no longer fits the 256 MiB Infinity Cache, so this is the honest HBM-roofline
    number for the gather-scale-segment-sum kernel.
(2) 100k-point radius graph with a dense blob (in-degree up to 32), F = 256.
Reports time, algorithmic GB/s (gather model) and compulsory GB/s (each row once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex  # noqa: E402
from kbench import timeit  # noqa: E402


def report(name, g, n, e, f, dev):
    x = torch.randn(n, f, device=dev)
    y = torch.empty(n, f, device=dev)
    ms = timeit(lambda: ops.hop(g.fwd, x, out=y), 10)
    alg = e * (8 + 4 * f) + n * (4 * f + 4)
    comp = e * 8 + n * (8 * f + 4)
    deg = (g.fwd.ptr[1:] - g.fwd.ptr[:-1]).float()
    print(f"{name}: N={n} E={e} F={f} deg mean {deg.mean():.1f} max {int(deg.max())}: {ms * 1e3:9.1f} us  "
          f"algorithmic {alg / ms / 1e6:8.0f} GB/s ({alg / ms / 1e6 / 8000:.2f} of 8 TB/s)  "
          f"compulsory {comp / ms / 1e6:8.0f} GB/s ({comp / ms / 1e6 / 8000:.2f})")
    msb = timeit(lambda: ops.hop(g.bwd, x, out=y, addend=y), 10)
    print(f"{'':{len(name)}}  transposed + addend: {msb * 1e3:9.1f} us  algorithmic "
          f"{(alg + n * 4 * f) / msb / 1e6:8.0f} GB/s")
    # bf16-stored features, fp32 accumulate (configs[4] / SURVEY 8(d) config 5)
    xb = x.bfloat16()
    for odt, ob in ((torch.bfloat16, 2), (torch.float32, 4)):
        yb = torch.empty(n, f, device=dev, dtype=odt)
        msh = timeit(lambda: ops.hop_bf16(g.fwd, xb, out=yb, out_dtype=odt), 10)
        algb = e * (8 + 2 * f) + n * (ob * f + 4)
        compb = e * 8 + n * ((2 + ob) * f + 4)
        print(f"{'':{len(name)}}  bf16 x -> {str(odt)[6:]:8s}: {msh * 1e3:9.1f} us  algorithmic "
              f"{algb / msh / 1e6:8.0f} GB/s ({algb / msh / 1e6 / 8000:.2f})  compulsory "
              f"{compb / msh / 1e6:8.0f} GB/s ({compb / msh / 1e6 / 8000:.2f})  "
              f"{e / msh / 1e6:7.2f} G edge-hops/s (fp32: {e / ms / 1e6:7.2f})")


def main():
    dev = torch.device("cuda:0")
    rest, _, _ = synth.make_batch(1)
    ei1 = rest.edge_index.numpy()
    copies = 2048
    n1 = rest.x.shape[0]
    ei = np.concatenate([ei1 + i * n1 for i in range(copies)], axis=1)
    n, e = n1 * copies, ei.shape[1]
    g = GraphIndex(torch.from_numpy(ei).to(dev), n)
    report("2048 soft meshes ", g, n, e, 256, dev)
    del g
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    g = GraphIndex(ei.to(dev), pos.shape[0])
    report("radius graph 100k", g, pos.shape[0], ei.shape[1], 256, dev)
    del g
    rest, rig, _ = synth.make_batch(32)
    g = GraphIndex(rest.edge_index.to(dev), rest.x.shape[0])
    report("everyday B=32 soft", g, rest.x.shape[0], rest.edge_index.shape[1], 256, dev)


if __name__ == "__main__":
    main()
