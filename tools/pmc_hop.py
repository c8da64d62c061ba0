#!/usr/bin/env python3
"""Run ONLY the launches that perform the F=256 hops of a B=32 step, in step order - what bench.py prices in its
`roofline` object: by default 4 `dc_hop_chain_f32` launches (forward and transposed 3-hop chain of each branch;
DC_HOP_CHAIN=0: 12 `dc_spmm_f32_rowmax` launches), with DC_MERGE_BRANCHES=1 (opt-in encoder path) 6 hop launches over
the merged adjacency - so that rocprofv3 --pmc passes can attribute HBM traffic to dc::k_hop_chain* / dc::k_spmm_wave.

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


#: kernels the hop launches of a step run on: the per-hop gather kernel and the fused 3-hop chain (dc_hopchain.hip)
HOP_KERNELS = ("k_spmm_wave", "k_hop_chain")


def hop_sequence(dev, parts, merged: bool, f: int = 256, segments=None):
    """The F=256 hop launches of ONE encoder step in step order on step-shaped slabs - what `roofline` prices
    and what the --pmc passes measure.  `parts` = [(edge_index, n), ...] (soft, rigid); `segments` = the batches'
    `Batch.segments()` (one per part, as the encoder passes them to `graph_index`): with them the default path runs
    every 3-hop chain as ONE `dc_hop_chain_f32` launch (4 launches per step), without them hop by hop (12).
    merged (opt-in encoder path): ONE adjacency over both graphs, 3 forward hops on the merged layer-2 slab + 3
    transposed hops on the merged gradient slab = 6 launches.
    -> (fn, launches, per-hop compulsory bytes of all hops, gather-model bytes of all hops)."""
    import torch
    from deformcontact_amd import ops
    from deformcontact_amd.graph import GraphIndex
    seq, comp, gath, launches = [], 0, 0, 0
    n_real = sum(n for _, n in parts)
    e_all = sum(int(ei.shape[1]) for ei, _ in parts)
    segments = segments or [None] * len(parts)
    graphs = [(GraphIndex.from_parts(parts), n_real, e_all)] if merged else \
        [(GraphIndex(ei, n, segments=sg), n, int(ei.shape[1])) for (ei, n), sg in zip(parts, segments)]
    for g, n, e in graphs:
        for bwd in (False, True):
            slab = ops._alloc_slab(g.num_nodes, 4 * f, dev).normal_()
            seq.append((g, slab, torch.zeros(g.num_nodes, device=dev), bwd))
            comp += 3 * (e * 8 + n * (8 * f + 4))                   # rows of padding nodes are not counted
            gath += 3 * (e * (8 + 4 * f) + n * (4 * f + 4))
            launches += 1 if ops.hop_chain_eligible(g, g.bwd if bwd else g.fwd, slab, f, 3) else 3

    def fn():
        for g, slab, rm, bwd in seq:             # forward / transposed chains of 3 hops + row maxima
            if bwd:
                ops.chained_hops(g, slab, f, 3, backward=False, rowmax=rm, transposed=True,
                                 rowmax_has_block0=True)
            else:
                ops.chained_hops(g, slab, f, 3, backward=False, rowmax=rm)
    return fn, launches, comp, gath


def run():
    import torch
    from deformcontact_amd import synth
    dev = torch.device("cuda:0")
    rest, _, rig = synth.make_batch(32)
    merged = os.environ.get("DC_MERGE_BRANCHES", "0") == "1"
    fn, _, _, _ = hop_sequence(dev, [(rest.edge_index.to(dev), rest.x.shape[0]),
                                     (rig.edge_index.to(dev), rig.x.shape[0])], merged,
                               segments=None if merged else [rest.segments(), rig.segments()])
    for _ in range(10):                          # the launches bench.py prices, in the same order
        fn()
    torch.cuda.synchronize()


def parse(fetch_dir, write_dir):
    def mean_counter(d, name):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if any(k in r["Kernel_Name"] for k in HOP_KERNELS) and r["Counter_Name"] == name]
        return sum(vals) / len(vals), len(vals)
    fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
    write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
    out = {"kernel": "the step's F=256 hop launches (dc::k_hop_chain<true,STEPS> / dc::k_spmm_wave<4,8,true>)",
           "launches_averaged": [nf, nw],
           "FETCH_SIZE_KiB_raw": round(fetch_kb, 1), "WRITE_SIZE_KiB": round(write_kb, 1),
           "correction": "FETCH_SIZE x2: gfx950 tallies 128-B requests of 16 B/lane reads at 64 B",
           "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
           "note": "B=32 working set (x 33.6 MB + slab) sits in the 256 MiB Infinity Cache; these are "
                   "L2 fabric-side request bytes (Infinity-Cache hits are counted, not excluded)"}
    print(json.dumps(out, indent=1))


def parse_l2(d):
    """Averages per k_spmm_wave launch of every counter in a rocprofv3 --pmc run of this script (e.g.
    TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum: L1 -> L2 read requests and L2 -> fabric read requests, 128 / 64-byte units
    as MI355X_MICROARCH.md states them)."""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in HOP_KERNELS):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            e = acc.setdefault(name, {})
            e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            e.setdefault("_dur_us_profiled", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(json.dumps({n: {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in e.items()}
                      for n, e in acc.items()}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--parse-l2":
        parse_l2(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
