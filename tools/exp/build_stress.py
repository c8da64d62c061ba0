#!/usr/bin/env python3
"""Is the one-launch segmented adjacency build bit-reproducible when two of them run side by side?  Soft batch on stream A,
rigid batch on stream B, caches cleared every time, every array of both adjacencies compared with the first build.
    python tools/exp/build_stress.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, clear_cache  # noqa: E402


def arrays(g):
    return [g.fwd.ptr, g.fwd.other, g.fwd.w, g.bwd.ptr, g.bwd.other, g.bwd.w]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda:0")
    bsz, sv, sr = int(os.environ.get("HUNT_B", "2")), int(os.environ.get("HUNT_SV", "256")), int(os.environ.get("HUNT_SR", "8"))
    rest, _, rig = (b.to(dev) for b in synth.make_batch(bsz, soft_vertices=sv, sphere_resolution=sr))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    want = None
    bad = 0
    names = ["fwd.ptr", "fwd.other", "fwd.w", "bwd.ptr", "bwd.other", "bwd.w"]
    for i in range(iters):
        clear_cache()
        with torch.cuda.stream(sa):
            ga = GraphIndex(rest.edge_index, rest.x.shape[0], segments=rest.segments())
        with torch.cuda.stream(sb):
            gb = GraphIndex(rig.edge_index, rig.x.shape[0], segments=rig.segments())
        torch.cuda.synchronize()
        cur = [t.clone() for t in arrays(ga) + arrays(gb)]
        if want is None:
            want = cur
            continue
        diff = [("soft." if j < 6 else "rigid.") + names[j % 6] for j, (a, b) in enumerate(zip(cur, want)) if not torch.equal(a, b)]
        if diff:
            bad += 1
            print(f"  iteration {i}: differ: {diff}", flush=True)
    print(f"{bad} of {iters - 1} concurrent builds differ from the first one")


if __name__ == "__main__":
    main()
