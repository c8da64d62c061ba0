"""CPU: host logic + the C-ABI library loads and exports everything include/*.h declares."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

import deformcontact_amd as dc
from deformcontact_amd import _lib, features
from deformcontact_amd.data import Batch, Data
from tests.helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    declared = _declared_symbols()
    assert len(declared) >= 6
    handle = ctypes.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/ but not exported"
    # the Python binding covers exactly the declared surface
    assert sorted(_lib.exported_names()) == declared
    assert _lib.lib().dc_version() >= 100


def test_abi_argument_errors_without_gpu():
    L = _lib.lib()
    assert L.dc_csr_workspace_bytes(-1, 4) < 0
    assert L.dc_csr_workspace_bytes(10, 4) > 0
    # bad arguments are rejected before any HIP call
    rc = L.dc_spmm_f32(None, None, None, None, 4, None, 0, None, 4, 3, 4, None)
    assert rc == -1 and b"null" in L.dc_last_error()
    rc = L.dc_csr_build(None, 5, 3, 2, 0, None, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"key_row" in L.dc_last_error()


def test_adjacency_build_plan_rule(monkeypatch):
    """dc_graph_build_plan: which pipeline an edge set of a given size takes (dc_csr.hip: bucket_plan) - host logic."""
    L = _lib.lib()
    for k in ("DC_CSR_BUCKETS", "DC_CSR_BUCKETS_MIN", "DC_CSR_BUCKET_SHIFT"):
        monkeypatch.delenv(k, raising=False)
    shift, nb = ctypes.c_int(-1), ctypes.c_int(-1)

    def plan(e, n, loops=0):
        rc = L.dc_graph_build_plan(e, n, loops, ctypes.byref(shift), ctypes.byref(nb))
        return rc, shift.value, nb.value
    assert plan(1_118_107, 100_000) == (1, 8, 391)             # BASELINE configs[4]: 256 nodes per bucket
    assert plan(196_224, 32_768)[0] == 0                        # the batch-32 soft graph: windowed pipeline
    assert plan(342_144, 57_344, 1)[0] == 0                     # ... merged with the rigid one, self loops appended
    assert plan((1 << 19) - 1, 1000)[0] == 0 and plan(1 << 19, 1000)[0] == 1
    rc, s, b = plan(1 << 20, 1 << 24)                           # many nodes: at most 2,048 per bucket, 8,192 buckets
    assert (rc, s, b) == (1, 11, 8192)
    assert plan(1 << 20, (1 << 24) + 1)[0] == 0                 # one bucket too many: windowed
    rc, s, b = plan(100_000_000, 4096)                          # very dense: the smallest buckets
    assert (rc, s, b) == (1, 4, 256)
    assert plan(0, 5)[0] == 0 and plan(5, 0)[0] == 0
    assert L.dc_graph_build_plan(-1, 5, 0, None, None) < 0
    monkeypatch.setenv("DC_CSR_BUCKETS", "1")
    assert plan(100, 50) == (1, 10, 1)                           # forced (tests of the kernels on small graphs)
    monkeypatch.setenv("DC_CSR_BUCKET_SHIFT", "6")
    assert plan(100, 50) == (1, 6, 1)
    monkeypatch.setenv("DC_CSR_BUCKETS", "0")
    assert plan(1_118_107, 100_000)[0] == 0


def test_no_cpu_fallback():
    conv = dc.nn.TAGConv(4, 8)
    x = torch.zeros(5, 4)
    ei = torch.zeros(2, 3, dtype=torch.long)
    with pytest.raises(RuntimeError, match="HIP device"):
        conv(x, ei)
    with pytest.raises(RuntimeError, match="HIP device"):
        dc.nn.GCNConv(4, 8)(x, ei)
    with pytest.raises(RuntimeError, match="HIP device"):
        dc.nn.GATConv(4, 8)(x, ei)


def test_product_never_imports_oracle():
    for path in glob.glob(os.path.join(ROOT, "deformcontact_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path


def test_state_dict_keys_match_reference():
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model
    m = load_model(EVERYDAY_NETWORK)
    sd = m.state_dict()
    assert sum(v.numel() for v in sd.values()) == 1_033_219
    for br, fin in (("resting", 21), ("rigid", 25)):
        for k in range(4):
            assert sd[f"conv_layers_{br}.0.lins.{k}.weight"].shape == (256, fin)
            assert sd[f"conv_layers_{br}.1.lins.{k}.weight"].shape == (256, 256)
        assert sd[f"conv_layers_{br}.1.bias"].shape == (256,)
    assert sd["multihead_attention.attention_heads.1.weight"].shape == (256, 256)
    assert sd["decoder.0.weight"].shape == (256, 768) and sd["decoder.9.weight"].shape == (3, 256)
    gat = dc.nn.GATConv(5, 7).state_dict()
    assert set(gat) == {"att_src", "att_dst", "bias", "lin.weight"}
    assert set(dc.nn.GCNConv(5, 7).state_dict()) == {"bias", "lin.weight"}


def test_to_log_freq_golden():
    z = load_golden("pos_encoding.npz")
    enc = features.to_log_freq(torch.from_numpy(z["pos"]), 3, 1)
    assert enc.shape == (37, 21)
    assert np.allclose(enc.numpy(), z["enc"], rtol=0, atol=1e-7)


class _M:
    def __init__(self, v, t):
        self.vertices, self.triangles = v, t


def test_mesh_to_graph_and_batch_golden():
    z = load_golden("mesh_graph_csr.npz")
    g1 = features.mesh_to_graph(_M(z["v1"], z["t1"]))
    g2 = features.mesh_to_graph(_M(z["v2"], z["t2"]))
    assert g1.edge_index.dtype == torch.int64 and g1.edge_index.is_contiguous()
    assert np.array_equal(g1.edge_index.numpy(), z["ei1"])
    assert np.array_equal(g2.edge_index.numpy(), z["ei2"])
    assert np.allclose(g1.x.numpy(), z["x1"], atol=1e-7)
    b = Batch.from_data_list([g1, g2, g1])
    assert np.array_equal(b.edge_index.numpy(), z["batch_edge_index"])     # bit-exact
    assert np.array_equal(b.ptr.numpy(), z["batch_ptr"])
    assert np.array_equal(b.batch.numpy(), z["batch_vec"])
    assert np.allclose(b.x.numpy(), z["batch_x"], atol=1e-7)
    assert b.num_graphs == 3 and len(b) == 3
    ex = b[1]
    assert np.array_equal(ex.edge_index.numpy(), z["ei2"]) and ex.x.shape == g2.x.shape
    c = b.clone()
    c.pos += 1.0
    assert not torch.equal(c.pos, b.pos)
    assert np.array_equal(c[2].edge_index.numpy(), z["ei1"])
    with pytest.raises(IndexError):
        b[3]


def test_batch_ragged_and_extra_attributes():
    d0 = Data(x=torch.zeros(3, 2), edge_index=torch.tensor([[0, 1], [1, 2]]), pos=torch.zeros(3, 3),
              force=torch.ones(3), name="a")
    d1 = Data(x=torch.ones(1, 2), edge_index=torch.zeros(2, 0, dtype=torch.long),
              pos=torch.ones(1, 3), force=torch.zeros(3), name="b")
    b = Batch.from_data_list([d0, d1])
    assert b.x.shape == (4, 2) and b.edge_index.shape == (2, 2)
    assert b.force.shape == (2, 3) and b.name == ["a", "b"]
    assert b[1].edge_index.shape == (2, 0) and b[1].name == "b"
    with pytest.raises(ValueError):
        Batch.from_data_list([])


def test_feature_rigid_layout():
    enc = torch.arange(42, dtype=torch.float32).reshape(2, 21)
    f = features.feature_rigid(torch.tensor([1.0, 2.0, 3.0]), 0.5, enc)
    assert f.shape == (2, 25)
    assert torch.equal(f[:, :3], torch.tensor([[1.0, 2.0, 3.0]] * 2))
    assert torch.equal(f[:, 3], torch.tensor([0.5, 0.5])) and torch.equal(f[:, 4:], enc)


def test_feature_rigid_matches_reference_function():
    """a10: ``features.feature_rigid`` against the output of the REFERENCE's own ``_feature_rigid``
    (``/root/reference/loaders/common.py:6-19``, imported by ``oracle/make_golden.py`` behind a stub
    ``open3d``) - bit for bit, also for a single-vertex encoding and a zero force."""
    z = load_golden("feature_rigid.npz")
    for i in range(int(z["n_cases"])):
        out = features.feature_rigid(torch.from_numpy(z[f"force_vector{i}"]), float(z[f"force{i}"]),
                                     torch.from_numpy(z[f"pos_enc{i}"]))
        assert out.dtype == torch.float32 and np.array_equal(out.numpy(), z[f"features{i}"])


def test_gcl_loss_golden():
    from deformcontact_amd.graphnet import gradient_consistency_loss
    from tests.helpers import G
    z = load_golden("gcl_loss.npz")
    ei = torch.from_numpy(z["edge_index"])
    p = G(None, ei, torch.from_numpy(z["pred_pos"]).requires_grad_(True))
    t = G(None, ei, torch.from_numpy(z["tgt_pos"]))
    loss = gradient_consistency_loss(p, t)
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    assert np.allclose(p.pos.grad.numpy(), z["grad_pred"], rtol=1e-5, atol=1e-8)


def test_pyg_alias_installs_modules():
    import sys
    dc.install_as_torch_geometric()
    from torch_geometric.nn import GATConv, GCNConv, TAGConv, knn  # noqa: F401
    from torch_geometric.data import Batch as B2, Data as D2
    assert TAGConv is dc.nn.TAGConv and B2 is Batch and D2 is Data
    for k in ("torch_geometric", "torch_geometric.nn", "torch_geometric.data"):
        sys.modules.pop(k, None)


def test_synth_everyday_shape():
    from deformcontact_amd import synth
    rest, deff, rig = synth.make_batch(2)
    assert rest.x.shape == (2048, 21) and rest.edge_index.shape == (2, 12264)
    assert rig.x.shape == (1524, 25) and rig.edge_index.shape == (2, 9120)
    deg = np.bincount(rig.edge_index.numpy()[1][:4560], minlength=762)
    assert sorted(np.unique(deg).tolist()) == [5, 6, 40]
    assert deff.pos.shape == rest.pos.shape


def test_prefetch_loader_matches_plain_iteration_cpu():
    """PrefetchLoader (worker thread, depth-2 queue) yields the batches of iterate_batches + to_batches
    in the same order with the same contents; worker exceptions surface in the consumer."""
    import torch
    from deformcontact_amd.loaders import (PrefetchLoader, SyntheticEverydayDataset, iterate_batches,
                                           to_batches)
    ds = SyntheticEverydayDataset(5, first_idx=3, soft_vertices=64, sphere_resolution=4)
    plain = [(c, to_batches(c, None)) for c in iterate_batches(ds, 2, shuffle=True, seed=7)]
    for dev in (None, "cpu"):
        got = list(PrefetchLoader(ds, 2, dev, shuffle=True, seed=7))
        assert len(got) == len(plain) == 3
        for (c0, b0), (c1, b1) in zip(plain, got):
            assert c0[0] == c1[0]
            for x, y in zip(b0, b1):
                assert torch.equal(x.x, y.x) and torch.equal(x.edge_index, y.edge_index)
                assert torch.equal(x.pos, y.pos) and torch.equal(x.batch, y.batch)

    # the same samples held in memory: identical batches, __getitem__ a lookup
    from deformcontact_amd.loaders import InMemoryDataset
    mem = InMemoryDataset(ds)
    assert len(mem) == len(ds)
    for (c0, b0), c1 in zip(plain, iterate_batches(mem, 2, shuffle=True, seed=7)):
        assert c0[0] == c1[0]
        for x, y in zip(b0, to_batches(c1, None)):
            assert torch.equal(x.x, y.x) and torch.equal(x.edge_index, y.edge_index)

    # a consumer that stops early leaves no worker thread behind (it would sit in queue.put forever)
    import threading
    import time
    before = threading.active_count()
    for i, _ in enumerate(PrefetchLoader(mem, 1, None, depth=1)):
        if i == 1:
            break
    for _ in range(50):
        if threading.active_count() == before:
            break
        time.sleep(0.02)
    assert threading.active_count() == before

    class Broken(SyntheticEverydayDataset):
        def __getitem__(self, i):
            if i == 3:
                raise RuntimeError("boom")
            return super().__getitem__(i)
    import pytest
    with pytest.raises(RuntimeError, match="boom"):
        list(PrefetchLoader(Broken(5, soft_vertices=64, sphere_resolution=4), 2, None))


def test_grad_bucket_views_are_16_byte_aligned():
    """An odd-sized parameter (the decoder's 3-element bias) must not misalign the gradient views behind it:
    the dense kernels need 16-byte aligned operands for their vector forms."""
    import torch
    from deformcontact_amd import dp
    params = [torch.nn.Parameter(torch.randn(*s)) for s in ((256, 3), (3,), (256, 256), (7,), (5, 5), (256,))]
    bucket = dp.GradBucket(params)
    offs = dp.flat_offsets(params)
    assert offs == [0, 768, 772, 772 + 65536, 772 + 65536 + 8, 772 + 65536 + 8 + 28, 772 + 65536 + 8 + 28 + 256]
    assert bucket.numel == offs[-1] and all(o % 4 == 0 for o in offs)
    for p, off in zip(params, offs):
        assert p.grad.shape == p.shape and p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * off
        assert p.grad.data_ptr() % 16 == bucket.flat.data_ptr() % 16
        p.grad.fill_(1.0)
    used = sum(p.numel() for p in params)
    assert float(bucket.flat.sum()) == used          # the padding elements stay zero
    bucket.zero()
    assert not bucket.flat.any() and all(bucket.owns(p, p.grad) for p in params)


def test_batch_records_its_layout_and_drops_it_when_edge_index_is_replaced():
    """``Batch.segments()``: host-side node / edge offsets of the collated graphs (what the one-launch adjacency build
    consumes); survives ``clone()``, is dropped when ``edge_index`` is assigned, can be re-declared."""
    import torch
    from deformcontact_amd.data import Batch, Data
    gs = [Data(x=torch.zeros(n, 2), edge_index=torch.randint(0, n, (2, e))) for n, e in ((5, 7), (3, 0), (4, 9))]
    b = Batch.from_data_list(gs)
    assert b.segments() == ((0, 5, 8, 12), (0, 7, 7, 16))
    assert b.clone().segments() == b.segments()
    seg = b.segments()
    b.edge_index = b.edge_index.clone()
    assert b.segments() is None
    b.assume_segments(seg)
    assert b.segments() == seg
    # every edge of graph i stays inside graph i's node range (the property the build relies on)
    n, e = seg
    for i in range(3):
        part = b.edge_index[:, e[i]:e[i + 1]]
        assert part.numel() == 0 or (int(part.min()) >= n[i] and int(part.max()) < n[i + 1])


def test_segment_layout_validation_is_host_side():
    """``graph._segment_arrays``: layouts that do not cover [0, N] / [0, E], descend, or exceed the per-graph caps are
    refused (the caller then runs the global pipeline) - decided on host data, before anything is launched."""
    import torch
    from deformcontact_amd import graph
    dev = torch.device("cpu")
    ok = graph._segment_arrays(((0, 5, 8), (0, 7, 16)), 8, 16, dev)
    assert ok is not None and ok[2] == 2 and list(ok[0]) == [0, 5, 8] and list(ok[1]) == [0, 7, 16]
    for bad in (((0, 5, 8), (0, 7, 15)), ((0, 5, 9), (0, 7, 16)), ((1, 5, 8), (0, 7, 16)), ((0, 5), (0, 7, 16)),
                ((0, 6, 5, 8), (0, 7, 9, 16)), ((0, 8), (0,))):
        assert graph._segment_arrays(bad[:2], 8, 16, dev) is None
    caps = graph._segment_arrays(((0, graph.SEG_MAX_NODES + 1), (0, 10)), graph.SEG_MAX_NODES + 1, 10, dev)
    assert caps is None
    assert graph._segment_arrays(((0, 4), (0, graph.SEG_MAX_EDGES + 1)), 4, graph.SEG_MAX_EDGES + 1, dev) is None
    assert graph._segment_arrays(((0, 4), (0, 0)), 4, 0, dev) is None          # no edges: nothing to sort
    # a graph with edges but no node: the one-launch build would index uninitialised LDS counters
    assert graph._segment_arrays(((0, 0, 8), (0, 3, 16)), 8, 16, dev) is None
    assert graph._segment_arrays(((0, 0, 8), (0, 0, 16)), 8, 16, dev) is not None       # an EMPTY graph is fine


def test_edge_equality_marks_do_not_outlive_the_edges_they_describe():
    """``train.losses`` takes the fused loss kernel only when the deformed batch's edges are KNOWN (on the host) to equal
    the rest batch's (``loaders.mark_edge_equality``, ``Batch.clone``); replacing or rewriting ``edge_index`` must drop
    that knowledge - the reference's ``models/losses.py:12-13`` uses each batch's own edges."""
    import torch
    from deformcontact_amd import loaders, train
    from deformcontact_amd.data import Batch, Data
    gs = [Data(x=torch.zeros(n, 2), pos=torch.zeros(n, 3), edge_index=torch.randint(0, n, (2, 6))) for n in (5, 4)]
    rest, deff = Batch.from_data_list(gs), Batch.from_data_list([g.clone() for g in gs])
    loaders.mark_edge_equality(rest, deff)
    assert train._edges_known_equal(deff, rest.edge_index)
    pred = rest.clone()
    assert train._edges_known_equal(pred, rest.edge_index)
    deff.edge_index = deff.edge_index.flip(1)                    # re-meshed: the mark goes with the old tensor
    assert not train._edges_known_equal(deff, rest.edge_index)
    pred.edge_index = pred.edge_index.clone()
    assert "_dc_cloned_edges" not in pred.__dict__ and not train._edges_known_equal(pred, rest.edge_index)
    deff2 = Batch.from_data_list([g.clone() for g in gs])
    loaders.mark_edge_equality(rest, deff2)
    deff2.edge_index.add_(0)                                     # rewritten in place: the version the mark names is gone
    assert not train._edges_known_equal(deff2, rest.edge_index)


def test_optional_per_graph_attention_mask_equals_a_block_diagonal_softmax():
    """SURVEY.md 8(f) rank 1 (optional per-graph block mask, OFF by default for parity): with the option on, every soft node
    attends only to the rigid nodes of its own sample - equal to the reference formula with -inf outside the diagonal
    blocks, forward and backward, for equal-size and for ragged batches; with the option off nothing changes."""
    import torch
    from deformcontact_amd.graphnet import CrossAttention
    torch.manual_seed(0)
    for ns, nr in (((0, 5, 10, 15), (0, 4, 8, 12)), ((0, 3, 9, 9, 14), (0, 2, 3, 7, 11))):
        att = CrossAttention(8, 2)
        xs = torch.randn(ns[-1], 8, requires_grad=True)
        xr = torch.randn(nr[-1], 8, requires_grad=True)
        plain = att(xs, xr)
        att.per_graph_mask = True
        got = att(xs, xr, ((ns, None), (nr, None)))
        mask = torch.full((ns[-1], nr[-1]), float("-inf"))
        for i in range(len(ns) - 1):
            mask[ns[i]:ns[i + 1], nr[i]:nr[i + 1]] = 0.0
        want = []
        for head in att.attention_heads:
            sc = head(xs) @ head(xr).t() + mask
            want.append(torch.softmax(sc, dim=-1) @ xr)
        want = torch.cat(want, dim=-1)
        assert torch.allclose(got, want, atol=1e-6)
        assert not torch.allclose(got, plain, atol=1e-3)
        g = torch.randn_like(got)
        ga = torch.autograd.grad(got, [xs, xr] + list(att.parameters()), g, retain_graph=True)
        gb = torch.autograd.grad(want, [xs, xr] + list(att.parameters()), g)
        for a, b in zip(ga, gb):
            assert torch.allclose(a, b, atol=1e-5)
        with pytest.raises(ValueError):
            att(xs, xr)
    # a sample without rigid nodes pools zeros
    att = CrossAttention(8, 1)
    att.per_graph_mask = True
    out = att(torch.randn(6, 8), torch.randn(4, 8), (((0, 3, 6), None), ((0, 4, 4), None)))
    assert out.shape == (6, 8) and float(out[3:].abs().max()) == 0.0 and float(out[:3].abs().max()) > 0


def test_chain_workgroups_request_the_whole_lds():
    """Round 5's fix of the chain launch's rare run-to-run difference: every `dc_hop_chain_f32` workgroup owns the whole LDS of
    its compute unit, so that no LDS-using workgroup of another kernel can be resident beside it (35 of 2,000 two-stream train
    steps differed with shared CUs and free LDS, 0 of 3,000 with this request; profiles/r05/README.md).  Pinned here so that a
    later "optimisation" of the request does not bring the condition back unnoticed."""
    from deformcontact_amd import _lib
    assert _lib.lib().dc_hop_chain_lds_request() == 160 * 1024
    assert _lib.lib().dc_hop_chain_max_nodes() == 4096


def test_batch_layout_travels_on_the_edge_index_tensor():
    """The reference hands a conv nothing but `graph.edge_index` (models/model.py:71,77): the batch layout that selects
    the one-launch adjacency build and the chain kernel is attached to that tensor by `Batch.from_data_list`, follows
    `.to()` / `.clone()`, dies with an in-place write (version counter) or a replaced tensor, and is sticky after
    `assume_segments` (static input buffers of a captured step)."""
    from deformcontact_amd.graph import edge_layout
    ds = [Data(x=torch.zeros(n, 3), edge_index=torch.tensor([[0, 1], [1, 0]]), pos=torch.zeros(n, 3)) for n in (2, 3, 4)]
    b = Batch.from_data_list(ds)
    seg = ((0, 2, 5, 9), (0, 2, 4, 6))
    assert b.segments() == seg and edge_layout(b.edge_index) == seg
    moved = b.to("cpu")
    assert edge_layout(moved.edge_index) == seg
    c = b.clone()
    assert c.edge_index is not b.edge_index and edge_layout(c.edge_index) == seg
    c.edge_index.add_(0)                                       # written in place: the tag no longer describes it
    assert edge_layout(c.edge_index) is None and edge_layout(b.edge_index) == seg
    c.assume_segments(seg)                                     # the caller's promise for every later content
    c.edge_index.add_(0)
    assert edge_layout(c.edge_index) == seg
    b.edge_index = torch.tensor([[0, 4], [4, 0]])              # replaced (radius graph, re-meshing): no layout
    assert b.segments() is None and edge_layout(b.edge_index) is None
    assert edge_layout(torch.zeros(2, 3, dtype=torch.long)) is None


def test_deferred_activation_semantics_cpu():
    """`deferred.DeferredActivation` (what a plain `conv(x, edge_index)` call returns): F.relu / torch.relu / .relu()
    run the layer once with the activation fused and return that plain tensor; anything else runs it without and applies
    the operation; metadata costs nothing; the value is an ordinary autograd tensor."""
    import torch.nn.functional as F
    from deformcontact_amd.deferred import DeferredActivation, deferred
    w = torch.randn(4, 3, requires_grad=True)
    x = torch.randn(5, 3)
    calls = []

    def run(relu):
        calls.append(relu)
        y = x @ w.t()
        return torch.relu(y) if relu else y
    d = deferred(run, 5, 4, x, True)
    assert isinstance(d, torch.Tensor) and isinstance(d, DeferredActivation)
    assert (tuple(d.shape), d.size(0), d.dim(), d.dtype, d.device, d.requires_grad, len(d), d.numel()) == \
        ((5, 4), 5, 2, torch.float32, x.device, True, 5, 20) and calls == []
    r = F.relu(d)
    assert type(r) is torch.Tensor and calls == [True] and torch.equal(r, torch.relu(x @ w.t()))
    assert F.dropout(r, p=0.0, training=True) is r             # models/model.py:72: the same tensor object goes on
    assert F.relu(d) is r and calls == [True]                  # computed once
    for use in (lambda t: t.sum(), lambda t: t + 1, lambda t: torch.cat([t, t], -1), lambda t: t.detach(),
                lambda t: t.data_ptr(), lambda t: t.is_contiguous(), lambda t: t[1:3], lambda t: repr(t)):
        calls.clear()
        use(deferred(run, 5, 4, x, True))
        assert calls == [False], use
    for act in (torch.relu, lambda t: t.relu(), lambda t: F.relu(t, inplace=True), lambda t: t.relu_()):
        calls.clear()
        assert type(act(deferred(run, 5, 4, x, True))) is torch.Tensor and calls == [True]
    deferred(run, 5, 4, x, True).relu().sum().backward()
    assert w.grad is not None and float(w.grad.abs().sum()) > 0
    with pytest.raises(RuntimeError):                          # the two autograd entry points that do not dispatch
        torch.autograd.backward([deferred(run, 5, 4, x, True)], [torch.ones(5, 4)])
    deferred(run, 5, 4, x, True).backward(torch.ones(5, 4))    # Tensor.backward does
    # an input written in place between the call and the first use: an eager conv would have used the old content
    late = deferred(run, 5, 4, x, True).guard(x, w)
    x.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        late.sum()


def test_plain_conv_call_checks_its_inputs_at_the_call_site():
    """Deferral postpones the launches, not the errors: CPU tensors / wrong shapes raise from `conv(x, edge_index)` itself."""
    conv = dc.nn.TAGConv(3, 8)
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(RuntimeError, match="HIP device"):
        conv(torch.zeros(2, 3), ei)
    for cls in (dc.nn.GCNConv, dc.nn.GATConv):
        with pytest.raises(RuntimeError, match="HIP device"):
            cls(3, 8)(torch.zeros(2, 3), ei)
