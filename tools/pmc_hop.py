#!/usr/bin/env python3
"""Run ONLY the F=256 hop launches bench.py prices in its `roofline` object (the 12 hop launches
of a B=32 step in step order: forward and transposed chains of both branches), so rocprofv3 --pmc
passes can attribute HBM traffic to dc::k_spmm_wave.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 tools/pmc_hop.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- python3 tools/pmc_hop.py
    python tools/pmc_hop.py --parse out/fetch out/write > profiles/pmc_hop.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a
wide (16 B/lane) streaming read (MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from deformcontact_amd import ops, synth
    from deformcontact_amd.graph import GraphIndex
    dev = torch.device("cuda:0")
    rest, _, rig = synth.make_batch(32)
    f = 256
    seq = []
    for b in (rest, rig):
        n = b.x.shape[0]
        g = GraphIndex(b.edge_index.to(dev), n)
        for bwd in (False, True):
            seq.append((g, ops._alloc_slab(n, 4 * f, dev).normal_(), torch.zeros(n, device=dev), bwd))
    for _ in range(10):                          # the launches bench.py prices, in the same order:
        for g, slab, rm, bwd in seq:             # forward / transposed chains of 3 hops + row maxima
            if bwd:
                ops.chained_hops(g, slab, f, 3, backward=False, rowmax=rm, transposed=True,
                                 rowmax_has_block0=True)
            else:
                ops.chained_hops(g, slab, f, 3, backward=False, rowmax=rm)
    torch.cuda.synchronize()


def parse(fetch_dir, write_dir):
    def mean_counter(d, name):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if "k_spmm_wave" in r["Kernel_Name"] and r["Counter_Name"] == name]
        return sum(vals) / len(vals), len(vals)
    fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
    write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
    out = {"kernel": "dc::k_spmm_wave<4,8,true>", "launches_averaged": [nf, nw],
           "FETCH_SIZE_KiB_raw": round(fetch_kb, 1), "WRITE_SIZE_KiB": round(write_kb, 1),
           "correction": "FETCH_SIZE x2: gfx950 tallies 128-B requests of 16 B/lane reads at 64 B",
           "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
           "note": "B=32 working set (x 33.6 MB + slab) sits in the 256 MiB Infinity Cache; these are "
                   "L2 fabric-side request bytes (Infinity-Cache hits are counted, not excluded)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
