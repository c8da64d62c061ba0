"""``dc_hop_chain_f32``: the K chained hops of a TAGConv layer (PyG ``tag_conv.py``: K dependent ``propagate`` calls,
reached from /root/reference/models/model.py:71,77) over a batch with a known layout as ONE launch with every graph's
slice resident in LDS.  The bar is bit-identity with K ``dc_spmm_f32_rowmax`` launches - blocks AND row maxima - on
mesh batches, ragged / empty / at-the-cap graphs, rows with more than 8 / 16 / 40 neighbours, the last rows of the
edge arrays (range-checked 16-byte loads), both directions, with and without weights; plus the C oracle on one case."""
import ctypes

import numpy as np
import pytest
import torch

from deformcontact_amd import _lib, ops, synth
from deformcontact_amd.graph import GraphIndex, current_stream_ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def batch_of_graphs(sizes, degs, seed, hub=None):
    """A block-diagonal batch: graph i has sizes[i] nodes and ~degs[i] in-edges per node (duplicates, self loops and
    isolated nodes included); ``hub`` = (graph, in-degree) gives node 3 of that graph a long segment."""
    rng = np.random.default_rng(seed)
    srcs, dsts, nptr, eptr = [], [], [0], [0]
    for i, (n, d) in enumerate(zip(sizes, degs)):
        e = n * d
        s = rng.integers(0, max(n, 1), e)
        t = rng.integers(0, max(n - n // 8, 1), e)          # the last nodes of a graph are never a target
        if hub is not None and hub[0] == i and n > 3:
            s = np.concatenate([s, rng.integers(0, n, hub[1])])
            t = np.concatenate([t, np.full(hub[1], 3)])
        if len(s) >= 12:
            s[:4], t[:4] = s[4:8], t[4:8]                   # duplicate edges
            s[8:12] = t[8:12]                               # self loops
        srcs.append(s + nptr[-1]), dsts.append(t + nptr[-1])
        nptr.append(nptr[-1] + n), eptr.append(eptr[-1] + len(s))
    ei = np.stack([np.concatenate(srcs), np.concatenate(dsts)]).astype(np.int64)
    return torch.from_numpy(ei), (tuple(nptr), tuple(eptr))


def reference_chain(adj, slab, f, k, rowmax, mode, weighted=True, src=0, direction=1):
    for j in range(k):
        a, b = src + j * direction, src + (j + 1) * direction
        ops.hop(adj, slab[:, a * f:(a + 1) * f], out=slab[:, b * f:(b + 1) * f], weighted=weighted, rowmax=rowmax,
                rowmax_mode=(mode if j == 0 else 2) if rowmax is not None else 0)


def run_both(g, adj, n, f, k, mode, weighted=True, src=0, direction=1, with_rowmax=True, seed=0, gcn=True):
    """``gcn``: let the kernel re-form the gcn_norm weights from its LDS degree table (the default for the adjacencies
    ``GraphIndex`` builds) instead of loading ``w`` - the other kernel behind the same entry point."""
    torch.manual_seed(seed)
    nblk = k + 1
    a = ops._alloc_slab(n, nblk * f, DEV)
    a.normal_()
    a[:, src * f:(src + 1) * f] *= torch.logspace(-3, 3, n, device=DEV)[:, None]     # rows of very different scales
    b = a.clone()
    ra = rb = None
    if with_rowmax:
        ra = torch.rand(n, device=DEV) * (10.0 if mode & 2 else 0.0)                 # the stored maxima mode 2 joins with
        rb = ra.clone()
    reference_chain(adj, a, f, k, ra, mode, weighted, src, direction)
    assert ops.hop_chain_eligible(g, adj, b, f, k)
    keep, ops.HOP_CHAIN_GCN = ops.HOP_CHAIN_GCN, gcn
    try:
        ops.hop_chain(g, adj, b, f, k, weighted=weighted, rowmax=rb, rowmax_mode=mode, src_block=src, direction=direction)
    finally:
        ops.HOP_CHAIN_GCN = keep
    torch.cuda.synchronize()
    assert torch.equal(a, b), f"blocks differ: {(a != b).sum().item()} elements"
    if with_rowmax:
        assert torch.equal(ra, rb), f"row maxima differ in {(ra != rb).sum().item()} rows"
    return a


@pytest.mark.parametrize("gcn", [True, False])
@pytest.mark.parametrize("bwd", [False, True])
def test_config1_b32_chains_equal_three_hops_bitwise(bwd, gcn):
    rest, _, rig = synth.make_batch(8)
    for b in (rest, rig):
        g = GraphIndex(b.edge_index.to(DEV), b.x.shape[0], segments=b.segments())
        assert g._segments is not None
        run_both(g, g.bwd if bwd else g.fwd, b.x.shape[0], 256, 3, 2 if bwd else 1, gcn=gcn)


@pytest.mark.parametrize("sizes,degs,f,k,hub", [
    ((1024, 1, 0, 513, 1024, 129, 128, 127), (6, 0, 3, 7, 9, 5, 2, 11), 256, 3, (3, 70)),   # ragged, empty, at the cap
    ((300, 300, 300), (6, 6, 6), 64, 2, (1, 17)),
    ((64,) * 5, (4,) * 5, 32, 1, None),
    ((1000, 24), (12, 3), 128, 3, (0, 200)),
    ((700,) * 100, (5,) * 100, 32, 2, (99, 41)),                                             # > 96 graphs: two launches
    # beyond 1,024 nodes per graph: 16-column slices up to 2,048 nodes, 8-column ones up to 4,096;
    # the batch layout alone decides (these graphs are beyond the segmented BUILD's edge cap: global build pipeline)
    ((1500, 1025, 7), (6, 3, 2), 256, 3, (0, 90)),
    ((2048, 2047, 0, 1), (5, 2, 0, 0), 64, 2, (1, 33)),
    ((3000, 2049), (6, 4), 256, 3, (0, 19)),
    ((4096, 10, 4095), (4, 1, 7), 32, 3, (2, 300)),
])
def test_ragged_batches_hubs_and_caps(sizes, degs, f, k, hub):
    ei, segs = batch_of_graphs(sizes, degs, seed=len(sizes) + f, hub=hub)
    n = segs[0][-1]
    g = GraphIndex(ei.to(DEV), n, segments=segs)
    assert g._layout is not None and g._seg_max_nodes == max(sizes)
    for adj, mode in ((g.fwd, 1), (g.bwd, 2), (g.fwd, 0), (g.bwd, 3)):
        for gcn in (True, False):
            run_both(g, adj, n, f, k, mode, seed=mode, gcn=gcn)
    run_both(g, g.fwd, n, f, k, 0, with_rowmax=False)
    run_both(g, g.fwd, n, f, k, 1, weighted=False)
    run_both(g, g.bwd, n, f, k, 1, src=k, direction=-1)


def test_last_rows_of_the_edge_arrays_and_the_c_oracle():
    """The 16-byte id / weight loads of the last rows run past the end of the arrays: the range check must hand back
    the in-range ids (and nothing of the tail may be lost).  Checked against the scalar C hop (oracle/hop_ref.c)."""
    from oracle import hop_c
    sizes, degs = (37, 5, 131), (3, 1, 2)
    ei, segs = batch_of_graphs(sizes, degs, seed=5)
    n = segs[0][-1]
    g = GraphIndex(ei.to(DEV), n, segments=segs)
    run_both(g, g.fwd, n, 32, 2, 1, gcn=False)
    slab = run_both(g, g.fwd, n, 32, 2, 1)
    ref = slab[:, :32].cpu().numpy().copy()
    w = hop_c.gcn_norm(ei.numpy(), n)
    for j in range(2):
        ref = hop_c.hop(ei.numpy(), w, ref)
        assert np.array_equal(ref, slab[:, (j + 1) * 32:(j + 2) * 32].cpu().numpy())


def test_c_abi_rejects_what_it_cannot_run():
    L = _lib.lib()

    def setup(sizes):
        ei, segs = batch_of_graphs(sizes, (2, 2), seed=1)
        n = segs[0][-1]
        g = GraphIndex(ei.to(DEV), n, segments=segs)
        slab = ops._alloc_slab(n, 2 * 32, DEV).normal_()
        return g, slab, n, (ctypes.c_int64 * 3)(*segs[0])

    def call(g, slab, n, nptr, f=32, k=1, src=0, direction=1, deg=False, w=True):
        return L.dc_hop_chain_f32(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr() if w else None,
                                  g.fwd.ptr.data_ptr() if deg else None, g.fwd.other.numel(),
                                  nptr, 2, slab.data_ptr(), slab.stride(0), n, f, k, src, direction, None, 0,
                                  current_stream_ptr(slab.device))
    assert L.dc_hop_chain_max_nodes() == 4096
    g, slab, n, nptr = setup((4097, 10))
    assert not ops.hop_chain_eligible(g, g.fwd, slab, 32, 1)        # a graph beyond the cap: hop by hop
    assert call(g, slab, n, nptr) != 0 and b"4097 nodes" in L.dc_last_error()
    g, slab, n, nptr = setup((1024, 11))
    assert ops.hop_chain_eligible(g, g.fwd, slab, 32, 1) and call(g, slab, n, nptr) == 0
    assert call(g, slab, n, nptr, f=24) != 0                        # F % 32
    assert call(g, slab, n, nptr, k=2) != 0                         # block 2 is outside the 2-block slab
    assert call(g, slab, n, nptr, src=1, direction=-1) == 0 and call(g, slab, n, nptr, src=0, direction=-1) != 0
    assert call(g, slab, n, (ctypes.c_int64 * 3)(0, 1024, n - 1)) != 0     # offsets must cover [0, N]
    assert call(g, slab, n, nptr, deg=True) == 0 and call(g, slab, n, nptr, deg=True, w=False) != 0
    torch.cuda.synchronize()


def test_tagconv_layer_and_encoder_identical_with_and_without_the_chain(monkeypatch):
    """The layer (forward + backward through the transposed chain) and the whole encoder: same bits either way."""
    from deformcontact_amd.graph import clear_cache
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(4))
    torch.manual_seed(0)
    enc = ContactEncoder([rest.x.size(1), rig.x.size(1)], 256, 2).to(DEV)
    outs = []
    for chain in (True, False):
        monkeypatch.setattr(ops, "HOP_CHAIN", chain)
        clear_cache()
        for p in enc.parameters():
            p.grad = None
        a, b = enc(rest, rig)
        (a.square().sum() + b.sum()).backward()
        outs.append([a.detach().cpu().numpy(), b.detach().cpu().numpy()] + [p.grad.cpu().numpy() for p in enc.parameters()])
    for u, v in zip(*outs):
        assert np.array_equal(u, v)


def test_which_chain_kernel_runs_by_graph_size():
    """Graphs of up to 1,024 nodes run `k_hop_chain_gcn<STEPS>` (adjacency tables in LDS) - small ones too: round 4 kept
    graphs of up to 512 nodes on the id / weight loading form because of a rare run-to-run difference, which round 5 traced
    to LDS co-residency with other kernels and removed at its source (every chain workgroup owns the whole LDS) -, larger
    ones `k_hop_chain<true, STEPS, LPR>` with 16- / 8-column slices; by name, through the launch log."""
    for sv, want, banned in ((256, "k_hop_chain_gcn<2>", "k_hop_chain<"), (1024, "k_hop_chain_gcn<8>", "k_hop_chain<"),
                             (1500, "k_hop_chain<true, 6, 4>", "k_hop_chain_gcn"), (3000, "k_hop_chain<true, 6, 2>", "k_spmm")):
        rest, _, _ = synth.make_batch(4, soft_vertices=sv, sphere_resolution=8)
        g = GraphIndex(rest.edge_index.to(DEV), rest.x.shape[0], segments=rest.segments())
        slab = ops._alloc_slab(rest.x.shape[0], 4 * 256, DEV).normal_()
        rm = torch.zeros(rest.x.shape[0], device=DEV)
        _lib.kernel_trace(True)
        ops.chained_hops(g, slab, 256, 3, backward=False, rowmax=rm, rowmax_zeroed=True)
        counts = _lib.kernel_trace_counts()
        _lib.kernel_trace(False)
        assert any(want in k for k in counts) and not any(banned in k for k in counts), (sv, counts)
