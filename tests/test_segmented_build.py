"""One-launch adjacency build for batches with a known layout (``dc_graph_build_segmented``): the arrays must be
those of the global pipeline (``dc_graph_build``) BIT FOR BIT - ptr, other, perm and the gcn_norm weights of both
sides - and of the oracle's scalar C build.  The layout is what ``Batch.from_data_list`` records
(/root/reference/loaders/everyday.py:96 collates with torch_geometric's Batch; the encoder then calls
conv(x, edge_index) on the collated graph, /root/reference/models/model.py:69-78)."""
import numpy as np
import pytest
import torch

from deformcontact_amd import graph as dc_graph
from deformcontact_amd import synth
from deformcontact_amd.data import Batch, Data
from deformcontact_amd.graph import GraphIndex, clear_cache, graph_index
from deformcontact_amd.graphnet import ContactEncoder
from oracle import hop_c
from tests.helpers import random_multigraph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _batch(shapes, seed=0, hub=None):
    graphs = []
    for i, (n, e) in enumerate(shapes):
        ei = random_multigraph(n, e, seed + 31 * i) if n > 0 else np.zeros((2, 0), np.int64)
        if hub is not None and i == hub[0] and e > hub[1]:
            ei[1, :hub[1]] = min(3, n - 1)                       # a node with a very long in-list
            ei[0, e - hub[1]:] = min(5, n - 1)                   # ... and one with a very long out-list
        graphs.append(Data(x=torch.zeros(n, 2), edge_index=torch.from_numpy(ei)))
    return Batch.from_data_list(graphs).to(DEV)


def _same_arrays(a: GraphIndex, b: GraphIndex):
    e = a.num_input_edges
    for x, y in ((a.fwd, b.fwd), (a.bwd, b.bwd)):
        assert np.array_equal(_np(x.ptr), _np(y.ptr))
        assert np.array_equal(_np(x.other)[:e], _np(y.other)[:e])
        assert np.array_equal(_np(x.perm)[:e], _np(y.perm)[:e])
        assert np.array_equal(_np(x.w)[:e].view(np.int32), _np(y.w)[:e].view(np.int32))


@pytest.mark.parametrize("shapes,hub", [
    ([(300, 2000), (77, 500), (1024, 6132)], None),
    ([(1, 0), (5, 9), (2, 40)], None),                           # an edgeless graph, tiny graphs
    ([(40, 100)], None),
    ([(700, 9000), (0, 0), (4096, 16384), (33, 1000)], None),     # an empty graph; one graph exactly at the caps
    ([(500, 6000), (900, 12000)], (1, 3000)),                    # groups far beyond the rank-by-counting length
    ([(1024, 6132)] * 32, None),                                  # the B = 32 soft batch's shape
    ([(30 + i % 7, 100 + 3 * i) for i in range(200)], None),      # more graphs than one launch carries (96)
])
def test_segmented_build_equals_global_pipeline_bitwise(shapes, hub):
    b = _batch(shapes, hub=hub)
    n, ei = b.x.size(0), b.edge_index
    seg = b.segments()
    assert seg is not None and seg[0][-1] == n and seg[1][-1] == ei.size(1)
    gs = GraphIndex(ei, n, segments=seg)
    assert gs._segments is not None, "layout within the caps must take the segmented path"
    gs.validate()
    gg = GraphIndex(ei, n)
    assert gg._segments is None
    _same_arrays(gs, gg)
    ptr, other, perm = hop_c.csr_build(_np(ei), n, 1)
    assert np.array_equal(_np(gs.fwd.ptr), ptr) and np.array_equal(_np(gs.fwd.other)[:ei.size(1)], other)
    assert np.array_equal(_np(gs.fwd.perm)[:ei.size(1)], perm)
    # rebuild into the same buffers after the edge_index buffer was refilled (the captured-step pattern)
    b2 = _batch(shapes, seed=99, hub=hub)
    ei.copy_(b2.edge_index)
    gs.rebuild()
    _same_arrays(gs, GraphIndex(ei.clone(), n))


def test_layouts_beyond_the_caps_or_inconsistent_fall_back_to_the_global_pipeline():
    b = _batch([(4097, 100), (10, 10)])
    assert GraphIndex(b.edge_index, b.x.size(0), segments=b.segments())._segments is None
    b = _batch([(100, 16385), (10, 10)])
    assert GraphIndex(b.edge_index, b.x.size(0), segments=b.segments())._segments is None
    b = _batch([(100, 500), (10, 10)])
    n, e = b.x.size(0), b.edge_index.size(1)
    for bad in (((0, 100), (0, 500, e)), ((0, 100, n), (0, 500, e - 1)), ((0, 100, 90, n), (0, 500, 505, e)),
                ((5, 100, n), (0, 500, e))):
        g = GraphIndex(b.edge_index, n, segments=bad)
        assert g._segments is None
        g.validate()
    # GCNConv's self-loop variant keeps the global pipeline
    assert GraphIndex(b.edge_index, n, self_loops=True, segments=b.segments())._segments is None


def test_edge_that_leaves_its_graph_is_flagged():
    b = _batch([(100, 500), (50, 200)])
    n = b.x.size(0)
    seg = b.segments()
    b.edge_index[0, 10] = 120                                    # graph 0's edge now starts in graph 1
    g = GraphIndex(b.edge_index, n, segments=seg)
    assert g._segments is not None
    with pytest.raises(IndexError):
        g.validate()
    # the arrays stay in range (nothing downstream can read out of bounds)
    e = b.edge_index.size(1)
    for adj in (g.fwd, g.bwd):
        p, o = _np(adj.ptr), _np(adj.other)[:e]
        assert p[0] == 0 and p[-1] == e and np.all(np.diff(p) >= 0) and o.min() >= 0 and o.max() < n


def test_replacing_edge_index_drops_the_recorded_layout():
    b = _batch([(100, 500), (50, 200)])
    assert b.segments() is not None and b.clone().segments() == b.segments()
    b.edge_index = b.edge_index.flip(1).contiguous()
    assert b.segments() is None


@pytest.mark.parametrize("overlap", [False, True])
def test_encoder_uses_the_segmented_build_and_matches_the_global_pipeline_bitwise(overlap, monkeypatch):
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(4, first_idx=3))
    torch.manual_seed(0)
    enc = ContactEncoder([rest.x.size(1), rig.x.size(1)], 256, 2).to(DEV)
    enc.overlap_branches = overlap
    outs = []
    for on in (True, False):
        clear_cache()
        monkeypatch.setattr(dc_graph, "SEGMENTED_BUILD", on)
        for p in enc.parameters():
            p.grad = None
        hs, hr = enc(rest, rig)
        (hs.square().sum() + hr.sum()).backward()
        g = graph_index(rest.edge_index, rest.x.size(0))
        assert (g._segments is not None) == on
        outs.append([_np(hs), _np(hr)] + [_np(p.grad) for p in enc.parameters()])
    for a, c in zip(*outs):
        assert np.array_equal(a, c)


def test_unvalidated_segmented_builds_share_one_status_word():
    """A build nobody validates takes the device's shared status word (no fill launch at the head of every new batch's
    build inside the captured step); `validate()` later on re-runs the build with a word of its own and still reports."""
    b = _batch([(100, 500), (50, 200)])
    n, seg = b.x.size(0), b.segments()
    g1, g2 = GraphIndex(b.edge_index, n, segments=seg), GraphIndex(b.edge_index.clone(), n, segments=seg)
    assert g1._status_shared and g2._status_shared and g1._status.data_ptr() == g2._status.data_ptr()
    assert not GraphIndex(b.edge_index, n, segments=seg, validate=True)._status_shared
    assert not GraphIndex(b.edge_index, n)._status_shared                      # the global pipeline keeps a private word
    g1.validate()
    assert not g1._status_shared and g1._status.data_ptr() != g2._status.data_ptr()
    _same_arrays(g1, g2)
    bad = b.edge_index.clone()
    bad[1, 3] = 140
    g3 = GraphIndex(bad, n, segments=seg)
    assert g3._status_shared
    g2.validate()                                                              # g3's flag sits in the shared word: not g2's
    with pytest.raises(IndexError):
        g3.validate()
