"""The bucketed adjacency build (dc_csr.hip, k_bk_*: edge lists in arbitrary order - a relabelled radius graph -
partitioned by node bucket, one workgroup per bucket) against the windowed pipeline and against the oracle: the
same ptr / other / perm / w, bit for bit, through every one of its paths (one LDS pass per bucket, several passes
of a bucket, a single node larger than a pass, long groups sorted in LDS, appended self loops, invalid ids, one
side per call, several edge_index parts).  Replaces /root/reference/models/model.py:71,77 (the implicit ordering of
scatter_add_ and gcn_norm) like the pipeline it is checked against."""
import numpy as np
import pytest
import torch

from deformcontact_amd import _lib
from deformcontact_amd.graph import GraphIndex, clear_cache, current_stream_ptr
from oracle import hop_c
from tests.helpers import random_multigraph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _build(monkeypatch, buckets, shift, make):
    monkeypatch.setenv("DC_CSR_BUCKETS", "1" if buckets else "0")
    if shift is None:
        monkeypatch.delenv("DC_CSR_BUCKET_SHIFT", raising=False)
    else:
        monkeypatch.setenv("DC_CSR_BUCKET_SHIFT", str(shift))
    clear_cache()
    g = make()
    torch.cuda.synchronize()
    return g


def _same(a: GraphIndex, b: GraphIndex):
    assert a.num_edges == b.num_edges
    e = a.num_edges
    for x, y in ((a.fwd, b.fwd), (a.bwd, b.bwd)):
        assert np.array_equal(_np(x.ptr), _np(y.ptr))
        assert np.array_equal(_np(x.perm)[:e], _np(y.perm)[:e])
        assert np.array_equal(_np(x.other)[:e], _np(y.other)[:e])
        if x.w is not None:
            assert np.array_equal(_np(x.w)[:e].view(np.int32), _np(y.w)[:e].view(np.int32))


def _oracle(ei, n, g):
    e = ei.shape[1]
    for key_row, adj in ((1, g.fwd), (0, g.bwd)):
        ptr, other, perm = hop_c.csr_build(ei, n, key_row)
        assert np.array_equal(_np(adj.ptr), ptr)
        assert np.array_equal(_np(adj.perm)[:e], perm)
        assert np.array_equal(_np(adj.other)[:e], other)


def _graph(kind):
    if kind == "random":                                   # one pass per bucket
        return random_multigraph(5000, 60000, 3), 5000
    if kind == "dense_buckets":                            # ~70k slots per 2048-node bucket: several passes each
        return random_multigraph(3000, 100000, 4), 3000
    if kind == "hub":                                      # one node beyond a pass (in and out), long groups of every
        n, e = 4000, 90000                                 # length around the rank-by-counting limit
        ei = random_multigraph(n, e, 5)
        ei[1, :20000] = 7
        ei[0, 30000:52000] = 2049
        at = 52000
        for i, length in enumerate((47, 48, 49, 50, 64, 65, 127, 500, 4097, 9000)):
            ei[1, at:at + length] = 100 + i
            at += length
        return ei, n
    if kind == "tiny":
        return random_multigraph(9, 30, 6), 9
    if kind == "one_node":
        return np.zeros((2, 5), np.int64), 1
    if kind == "ragged_last_bucket":                       # N not a multiple of the bucket size; isolated tail nodes
        return random_multigraph(2049 + 17, 30000, 8), 2049 + 17
    raise KeyError(kind)


@pytest.mark.parametrize("kind,shift", [("random", None), ("random", 4), ("dense_buckets", 11), ("dense_buckets", None),
                                        ("hub", None), ("hub", 11), ("hub", 1), ("tiny", None), ("one_node", None),
                                        ("ragged_last_bucket", 11), ("ragged_last_bucket", 6)])
def test_bucketed_build_equals_windowed_pipeline_and_oracle_bitwise(kind, shift, monkeypatch):
    ei_np, n = _graph(kind)
    ei = torch.from_numpy(ei_np).to(DEV)
    ref = _build(monkeypatch, False, None, lambda: GraphIndex(ei, n, validate=True))
    got = _build(monkeypatch, True, shift, lambda: GraphIndex(ei, n, validate=True))
    _same(got, ref)
    _oracle(ei_np, n, got)


@pytest.mark.parametrize("shift", [None, 3])
def test_bucketed_build_with_appended_self_loops(shift, monkeypatch):
    n = 3000
    ei_np = random_multigraph(n, 40000, 9, self_loops=True)
    ei = torch.from_numpy(ei_np).to(DEV)
    ref = _build(monkeypatch, False, None, lambda: GraphIndex(ei, n, self_loops=True, normalize=True))
    got = _build(monkeypatch, True, shift, lambda: GraphIndex(ei, n, self_loops=True, normalize=True))
    assert got.num_edges == ref.num_edges == int(np.sum(ei_np[0] != ei_np[1])) + n
    _same(got, ref)


def test_bucketed_build_reports_an_out_of_range_id_and_skips_the_edge(monkeypatch):
    n = 500
    ei_np = random_multigraph(n, 6000, 10)
    ei_np[0, 77] = n + 3
    ei_np[1, 4000] = -1
    ei = torch.from_numpy(ei_np).to(DEV)
    with pytest.raises(IndexError):
        _build(monkeypatch, True, None, lambda: GraphIndex(ei, n, validate=True))
    ref = _build(monkeypatch, False, None, lambda: GraphIndex(ei, n))
    got = _build(monkeypatch, True, None, lambda: GraphIndex(ei, n))
    assert int(got._status) == int(ref._status) == 1
    assert np.array_equal(_np(got.fwd.ptr), _np(ref.fwd.ptr)) and int(got.fwd.ptr[-1]) == 6000 - 2
    m = 6000 - 2
    for x, y in ((got.fwd, ref.fwd), (got.bwd, ref.bwd)):
        assert np.array_equal(_np(x.perm)[:m], _np(y.perm)[:m]) and np.array_equal(_np(x.other)[:m], _np(y.other)[:m])


def test_bucketed_build_of_several_parts(monkeypatch):
    parts_np = [(random_multigraph(700, 9000, 11), 700), (random_multigraph(1300, 20000, 12), 1300),
                (np.zeros((2, 0), np.int64), 40)]
    parts = [(torch.from_numpy(ei).to(DEV), n) for ei, n in parts_np]
    ref = _build(monkeypatch, False, None, lambda: GraphIndex.from_parts(parts))
    got = _build(monkeypatch, True, 5, lambda: GraphIndex.from_parts(parts))
    _same(got, ref)


def test_single_side_entry_takes_the_bucketed_build(monkeypatch):
    L = _lib.lib()
    n, e = 2500, 50000
    ei_np = random_multigraph(n, e, 13)
    ei_np[1, :15000] = 3                                     # beyond one pass
    ei = torch.from_numpy(ei_np).to(DEV)
    ref = _build(monkeypatch, False, None, lambda: GraphIndex(ei, n))
    monkeypatch.setenv("DC_CSR_BUCKETS", "1")
    st = current_stream_ptr(torch.device(DEV))
    status = torch.ones(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(L.dc_csr_workspace_bytes(e, n), dtype=torch.uint8, device=DEV)
    outs = {}
    for key_row, deg in ((1, None), (0, "f")):
        ptr = torch.empty(n + 1, dtype=torch.int32, device=DEV)
        other, perm = (torch.empty(e, dtype=torch.int32, device=DEV) for _ in range(2))
        w = torch.empty(e, dtype=torch.float32, device=DEV)
        rc = L.dc_csr_build(ei.data_ptr(), e, n, key_row, 0, ptr.data_ptr(), other.data_ptr(), perm.data_ptr(),
                            outs["f"][0].data_ptr() if deg else None, w.data_ptr(), status.data_ptr(), ws.data_ptr(),
                            ws.numel(), st)
        assert rc == 0
        outs["f" if key_row else "b"] = (ptr, other, perm, w)
    assert int(status) == 0
    for name, adj in (("f", ref.fwd), ("b", ref.bwd)):
        ptr, other, perm, w = outs[name]
        assert torch.equal(ptr, adj.ptr) and torch.equal(other, adj.other)
        assert torch.equal(perm, adj.perm) and torch.equal(w, adj.w)


def test_large_shuffled_edge_lists_take_the_bucketed_kernels_by_default(monkeypatch):
    """the size rule: 2^19 slots and more (the relabelled 100k-point radius graph of BASELINE configs[4])"""
    monkeypatch.delenv("DC_CSR_BUCKETS", raising=False)
    monkeypatch.delenv("DC_CSR_BUCKET_SHIFT", raising=False)
    n, e = 60000, 600000
    ei_np = random_multigraph(n, e, 14)
    ei = torch.from_numpy(ei_np).to(DEV)
    clear_cache()
    _lib.kernel_trace(True)
    g = GraphIndex(ei, n, validate=True)
    _lib.kernel_trace(False)
    names = _lib.kernel_trace_counts()
    assert any("k_bk_build" in k for k in names), sorted(names)
    assert not any("k_fill" in k for k in names)
    _oracle(ei_np, n, g)
    small = torch.from_numpy(random_multigraph(3000, 30000, 15)).to(DEV)
    _lib.kernel_trace(True)
    GraphIndex(small, 3000, validate=True)
    _lib.kernel_trace(False)
    names = _lib.kernel_trace_counts()
    assert any("k_fill" in k for k in names) and not any("k_bk_" in k for k in names)
