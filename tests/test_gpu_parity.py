"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures.  Bit-exact for indices; <= 1e-5 rel (max|a-b| / max|ref|) for fp32 as
BASELINE.json's north_star states (the hop itself is checked bit-for-bit against the
scalar C restatement)."""
import numpy as np
import pytest
import torch

import deformcontact_amd as dc
from deformcontact_amd import ops
from deformcontact_amd.graph import GraphIndex, clear_cache
from oracle import hop_c, pyg_ref
from oracle.weights import fill_state_dict_, hashed_uniform
from tests.helpers import (G, record_parity, assert_parity, golden_graphs, load_golden, random_multigraph, rel_err,
                           row_rel_err)

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------- #
# topology: bit-exact
# --------------------------------------------------------------------------- #
def _check_csr(ei, n):
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n, validate=True)
    for key_row, adj in ((1, g.fwd), (0, g.bwd)):
        ptr, other, perm = hop_c.csr_build(ei, n, key_row)
        order = np.argsort(ei[key_row], kind="stable")
        assert np.array_equal(perm, order.astype(np.int32))
        e = ei.shape[1]
        assert np.array_equal(_np(adj.ptr), ptr)
        assert np.array_equal(_np(adj.perm)[:e], perm)
        assert np.array_equal(_np(adj.other)[:e], other)
    w_edge = hop_c.gcn_norm(ei, n)                    # edge order
    e = ei.shape[1]
    assert rel_err(_np(g.fwd.w)[:e], w_edge[_np(g.fwd.perm)[:e]]) < 1e-6
    assert rel_err(_np(g.bwd.w)[:e], w_edge[_np(g.bwd.perm)[:e]]) < 1e-6
    return g


def test_csr_golden_fixture_bit_exact():
    z = load_golden("mesh_graph_csr.npz")
    ei, n = z["batch_edge_index"], z["batch_x"].shape[0]
    g = _check_csr(ei, n)
    assert np.array_equal(_np(g.fwd.ptr), z["rowptr"])
    assert np.array_equal(_np(g.fwd.perm), z["perm"])
    assert np.array_equal(_np(g.fwd.other), z["src_sorted"])
    assert np.array_equal(_np(g.bwd.ptr), z["colptr"])
    assert np.array_equal(_np(g.bwd.perm), z["perm_t"])
    assert np.array_equal(_np(g.bwd.other), z["dst_sorted"])


@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (5, 0, 1), (1, 7, 2), (64, 64, 3), (1000, 20000, 4),
                                      (4099, 30011, 5), (70000, 300000, 6)])
def test_csr_random_multigraphs_bit_exact(n, e, seed):
    _check_csr(random_multigraph(n, e, seed), n)


def test_csr_hub_node_long_segment():
    n, e = 300, 40000
    ei = random_multigraph(n, e, 11)
    ei[1, : e // 2] = 7                                   # one destination with 20k in-edges
    _check_csr(ei, n)


@pytest.mark.parametrize("hub", [48, 49, 300, 4096, 4097, 9000, 70000])
def test_csr_hub_lengths_around_every_threshold(hub):
    """Groups up to 48 ids rank by counting, longer ones are sorted by one workgroup (LDS up to
    4096 ids, in place beyond): lengths on both sides of each switch, as destination AND as
    source hub, plus a second long group so several listed groups share the sorting blocks."""
    n = 500
    e = hub + 3000
    ei = random_multigraph(n, e, 100 + hub % 97)
    rng = np.random.default_rng(hub)
    pos = rng.permutation(e)[:hub]
    ei[1, pos] = 7                                        # hub as destination
    ei[1, (ei[1] == 7) & ~np.isin(np.arange(e), pos)] = 8
    pos2 = rng.permutation(e)[:hub]
    ei[0, pos2] = 11                                      # hub as source
    ei[0, (ei[0] == 11) & ~np.isin(np.arange(e), pos2)] = 12
    ei[1, rng.permutation(e)[:77]] = 5                    # a second listed group
    g = _check_csr(ei, n)
    deg_in, deg_out = np.bincount(ei[1], minlength=n), np.bincount(ei[0], minlength=n)
    assert deg_in[7] > hub - 100 and deg_out[11] > hub - 100          # the hubs survived the edits
    assert _np(g.fwd.ptr)[8] - _np(g.fwd.ptr)[7] == deg_in[7]
    assert _np(g.bwd.ptr)[12] - _np(g.bwd.ptr)[11] == deg_out[11]


def test_csr_single_side_entry_equals_pair_build():
    """dc_csr_build (one side per call, the round-1 entry point) against dc_graph_build."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    L = _lib.lib()
    n, e = 700, 9000
    ei_np = random_multigraph(n, e, 21)
    ei_np[1, :200] = 3
    ei = torch.from_numpy(ei_np).to(DEV)
    g = GraphIndex(ei, n)
    st = current_stream_ptr(torch.device(DEV))
    status = torch.ones(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(L.dc_csr_workspace_bytes(e, n), dtype=torch.uint8, device=DEV)
    outs = {}
    for key_row, deg in ((1, None), (0, "f")):
        ptr = torch.empty(n + 1, dtype=torch.int32, device=DEV)
        other, perm = (torch.empty(e, dtype=torch.int32, device=DEV) for _ in range(2))
        w = torch.empty(e, dtype=torch.float32, device=DEV)
        rc = L.dc_csr_build(ei.data_ptr(), e, n, key_row, 0, ptr.data_ptr(), other.data_ptr(),
                            perm.data_ptr(), outs["f"][0].data_ptr() if deg else None, w.data_ptr(),
                            status.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0
        outs["f" if key_row else "b"] = (ptr, other, perm, w)
    assert int(status) == 0                                  # SET by the call
    for name, adj in (("f", g.fwd), ("b", g.bwd)):
        ptr, other, perm, w = outs[name]
        assert torch.equal(ptr, adj.ptr) and torch.equal(other, adj.other)
        assert torch.equal(perm, adj.perm) and torch.equal(w, adj.w)


def test_graph_rebuild_in_place_and_inside_a_captured_graph():
    """A captured step contains the build of the edge_index buffer it reads: refill the buffer,
    replay, and the adjacency (and a hop through it) follow the new topology."""
    from deformcontact_amd.graph import graph_index
    n, e, f = 400, 3000, 32
    eis = [random_multigraph(n, e, 31 + i) for i in range(3)]
    buf = torch.from_numpy(eis[0]).to(DEV)
    x = torch.from_numpy(hashed_uniform((n, f), 1, 1.0)).to(DEV)
    clear_cache()
    g_eager = graph_index(buf, n)
    assert graph_index(buf, n) is g_eager                    # eager reuse
    # in-place rebuild (eager)
    buf.copy_(torch.from_numpy(eis[1]))
    g_eager.rebuild()
    ptr, other, perm = hop_c.csr_build(eis[1], n, 1)
    assert np.array_equal(_np(g_eager.fwd.ptr), ptr) and np.array_equal(_np(g_eager.fwd.other), other)
    # capture: the eagerly built entry must NOT be reused
    y = torch.empty(n, f, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.hop(graph_index(buf, n).fwd, x, out=y)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_cap = graph_index(buf, n)
        assert g_cap is not g_eager
        assert graph_index(buf, n) is g_cap                  # reused inside the same capture
        ops.hop(g_cap.fwd, x, out=y)
    assert graph_index(buf, n) is not g_cap                  # and not outside it
    for i in (2, 0, 1):
        buf.copy_(torch.from_numpy(eis[i]))
        graph.replay()
        torch.cuda.synchronize()
        w_edge = hop_c.gcn_norm(eis[i], n)
        assert np.array_equal(_np(y), hop_c.hop(eis[i], w_edge, _np(x))), i
    # opt-out for a topology the caller guarantees constant
    g2 = graph_index(buf, n)
    g2._static_ok = True
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        assert graph_index(buf, n) is g2
        ops.hop(g2.fwd, x, out=y)
    graph2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(_np(y), hop_c.hop(eis[1], hop_c.gcn_norm(eis[1], n), _np(x)))


def test_content_hash_distinguishes_order_and_values():
    from deformcontact_amd.graph import content_hash
    a = torch.from_numpy(random_multigraph(300, 5000, 41)).to(DEV)
    b = a.clone()
    assert content_hash(a) == content_hash(b)
    b[0, 17], b[0, 18] = a[0, 18].item(), a[0, 17].item()   # swap two entries
    assert (a[0, 17] != a[0, 18]) and content_hash(a) != content_hash(b)
    c = a.clone()
    c[1, 4999] += 1
    assert content_hash(a) != content_hash(c)
    assert content_hash(a[:, :0].contiguous()) == 0


def test_csr_self_loop_rewrite_matches_pyg_loop_utils():
    n = 50
    ei = random_multigraph(n, 400, 12, self_loops=True)
    t = torch.from_numpy(ei)
    ref = pyg_ref.add_remaining_self_loops(t, n).numpy()             # GCN
    ref2 = pyg_ref.add_self_loops(pyg_ref.remove_self_loops(t), n).numpy()   # GAT
    assert np.array_equal(ref, ref2)
    g = GraphIndex(t.to(DEV), n, self_loops=True, normalize=True)
    e2 = ref.shape[1]
    assert g.num_edges == e2
    order = np.argsort(ref[1], kind="stable")
    assert np.array_equal(_np(g.fwd.other)[:e2], ref[0][order].astype(np.int32))
    assert np.array_equal(_np(g.fwd.ptr), np.concatenate([[0], np.cumsum(np.bincount(ref[1], minlength=n))]))
    order_t = np.argsort(ref[0], kind="stable")
    assert np.array_equal(_np(g.bwd.other)[:e2], ref[1][order_t].astype(np.int32))
    _, w = pyg_ref.gcn_norm(t, n, add_loops=True)
    assert rel_err(_np(g.fwd.w)[:e2], w.numpy()[order]) < 1e-6


def test_out_of_range_edge_is_reported():
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]], dtype=torch.long, device=DEV)
    with pytest.raises(IndexError):
        GraphIndex(ei, 3, validate=True)


# --------------------------------------------------------------------------- #
# the hop: bit-exact against the scalar C restatement
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,e,f", [(57, 400, 21), (57, 400, 25), (300, 3000, 256), (300, 3000, 32),
                                   (129, 900, 64), (64, 500, 128), (33, 200, 1), (33, 200, 7),
                                   (40, 300, 260), (40, 300, 1024), (20, 90, 84), (500, 9000, 16)])
def test_hop_bit_exact_vs_c_oracle(n, e, f):
    ei = random_multigraph(n, e, n + f)
    x = hashed_uniform((n, f), f, 2.0)
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    y = ops.hop(g.fwd, torch.from_numpy(x).to(DEV))
    w = hop_c.gcn_norm(ei, n)
    # same w on both sides so the comparison isolates the gather/scale/segment-sum
    w_dev = np.zeros(e, np.float32)
    w_dev[_np(g.fwd.perm)[:e]] = _np(g.fwd.w)[:e]
    yc = hop_c.hop(ei, w_dev, x)
    assert np.array_equal(_np(y), yc), f"hop differs: rel {rel_err(_np(y), yc):.3e}"
    assert rel_err(w_dev, w) < 1e-6
    # transposed operator: <A x, u> == <x, A^T u>
    u = hashed_uniform((n, f), f + 1, 2.0)
    yt = ops.hop(g.bwd, torch.from_numpy(u).to(DEV))
    lhs = float((_np(y).astype(np.float64) * u).sum())
    rhs = float((x.astype(np.float64) * _np(yt)).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


def _bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("n,e,f", [(300, 3000, 256), (57, 400, 21), (129, 900, 64), (40, 300, 264),
                                   (33, 200, 1), (64, 700, 8), (500, 9000, 16), (40, 300, 1024)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_hop_bf16_storage_bit_exact_vs_c_oracle(n, e, f, out_f32):
    """bf16-stored features, fp32 accumulation; bit-exact (fp32 result,
    or the bf16 bit patterns after the single final rounding)."""
    ei = random_multigraph(n, e, n + f + 5)
    xb = torch.from_numpy(hashed_uniform((n, f), f + 3, 2.0)).to(DEV).bfloat16()
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    w_dev = np.zeros(e, np.float32)
    w_dev[_np(g.fwd.perm)[:e]] = _np(g.fwd.w)[:e]
    odt = torch.float32 if out_f32 else torch.bfloat16
    y = ops.hop_bf16(g.fwd, xb, out_dtype=odt)
    yc = hop_c.hop_bf16(ei, w_dev, _bf16_bits(xb), out_f32)
    got = _np(y) if out_f32 else _bf16_bits(y)
    assert np.array_equal(got, yc)
    # unweighted (w == NULL) and the fp32 result agrees with the fp32 hop on the widened rows
    y1 = ops.hop_bf16(g.fwd, xb, weighted=False, out_dtype=torch.float32)
    assert torch.equal(y1, ops.hop(g.fwd, xb.float(), weighted=False))


def test_hop_bf16_strided_views_addend_and_errors():
    n, e, f = 200, 1500, 256
    ei = random_multigraph(n, e, 11)
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    slab = torch.from_numpy(hashed_uniform((n, 3 * f), 9, 2.0)).to(DEV).bfloat16()
    x = slab[:, f:2 * f]
    ref32 = ops.hop_bf16(g.fwd, x.contiguous(), out_dtype=torch.float32)
    out = slab[:, 2 * f:]
    add = out.clone()
    ops.hop_bf16(g.fwd, x, out=out)                                  # strided bf16 in / out
    assert torch.equal(out, ref32.bfloat16())
    ops.hop_bf16(g.fwd, x, out=out, addend=add)                      # y = bf16(add + A x)
    add32 = add.float()
    exp = ops.hop(g.fwd, x.float(), addend=add32)                    # same fp32 running sum
    assert torch.equal(out, exp.bfloat16())
    o32 = ops.hop_bf16(g.fwd, x, addend=add32, out_dtype=torch.float32)
    assert torch.equal(o32, exp)
    with pytest.raises(ValueError):
        ops.hop_bf16(g.fwd, x.float())
    with pytest.raises(Exception):
        ops.hop_bf16(g.fwd, x, out=x)
    # isolated nodes / empty rows give exact zeros
    g0 = GraphIndex(torch.zeros((2, 0), dtype=torch.int64, device=DEV), 5)
    z = ops.hop_bf16(g0.fwd, torch.ones(5, 24, device=DEV).bfloat16())
    assert z.dtype == torch.bfloat16 and not z.any()


def test_hop_bf16_beyond_cache_row_pair_kernel_and_radius_graph():
    """The row-pair kernel that takes over once x exceeds 128 MiB (odd N: the last pair is half
    empty), bit-exact against the C oracle; and BASELINE configs[4]: the 100k-point radius graph
    with bf16 features equals the fp32 hop over the widened rows bit for bit."""
    n, f = 270_001, 256
    rng = np.random.default_rng(5)
    e = 6 * n
    src = rng.integers(0, n, e)
    dst = np.minimum(n - 1, (src + rng.integers(-40, 41, e)).clip(0))       # banded, like a mesh
    ei = np.stack([src, dst]).astype(np.int64)
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    xb = (torch.rand(n, f, device=DEV) * 2 - 1).bfloat16()
    assert xb.numel() * 2 > 128 << 20
    w_dev = np.zeros(e, np.float32)
    w_dev[_np(g.fwd.perm)[:e]] = _np(g.fwd.w)[:e]
    for out_f32 in (False, True):
        y = ops.hop_bf16(g.fwd, xb, out_dtype=torch.float32 if out_f32 else torch.bfloat16)
        yc = hop_c.hop_bf16(ei, w_dev, _bf16_bits(xb), out_f32)
        assert np.array_equal(_np(y) if out_f32 else _bf16_bits(y), yc)
    add = torch.rand(n, f, device=DEV)
    assert torch.equal(ops.hop_bf16(g.bwd, xb, addend=add, out_dtype=torch.float32),
                       ops.hop(g.bwd, xb.float(), addend=add))
    del g, xb, add, y
    from deformcontact_amd import synth
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    g = GraphIndex(ei.to(DEV), pos.shape[0])
    xb = (torch.rand(pos.shape[0], f, device=DEV) * 2 - 1).bfloat16()
    y32 = ops.hop_bf16(g.fwd, xb, out_dtype=torch.float32)
    assert torch.equal(y32, ops.hop(g.fwd, xb.float()))
    assert torch.equal(ops.hop_bf16(g.fwd, xb), y32.bfloat16())


def test_hop_strided_views_and_addend():
    n, e, f = 200, 1500, 256
    ei = random_multigraph(n, e, 3)
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    slab = torch.from_numpy(hashed_uniform((n, 3 * f), 8, 2.0)).to(DEV)
    x = slab[:, f:2 * f]
    ref = ops.hop(g.fwd, x.contiguous())
    out = slab[:, 2 * f:]
    add = out.clone()
    ops.hop(g.fwd, x, out=out)
    assert torch.equal(out, ref)
    ops.hop(g.fwd, x, out=out, addend=add)                       # y = add + A x
    exp = _np(add).astype(np.float64) + _np(ref)
    assert rel_err(_np(out), exp) < 1e-6
    with pytest.raises(Exception):
        ops.hop(g.fwd, x, out=x)


def test_empty_graph_and_isolated_nodes():
    n, f = 9, 256
    ei = torch.zeros(2, 0, dtype=torch.long, device=DEV)
    g = GraphIndex(ei, n)
    y = ops.hop(g.fwd, torch.ones(n, f, device=DEV))
    assert torch.count_nonzero(y) == 0
    conv = dc.nn.TAGConv(f, 8).to(DEV)
    x = torch.randn(n, f, device=DEV)
    ref = torch.nn.functional.linear(x, conv.lins[0].weight) + conv.bias
    assert rel_err(_np(conv(x, ei)), _np(ref)) < TOL


# --------------------------------------------------------------------------- #
# dense block (fp32 MFMA) through the C ABI vs float64
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,fi,fo,nseg,relu", [
    (1, 21, 256, 4, True), (63, 25, 256, 4, False), (65, 32, 32, 4, True), (200, 256, 256, 4, True),
    (130, 256, 130, 2, False), (64, 16, 3, 1, True), (257, 84, 260, 3, True), (1000, 256, 256, 1, False)])
@pytest.mark.parametrize("split", [0, 6, 3, 1])
def test_dense_block_fwd_bwd_vs_float64(n, fi, fo, nseg, relu, split):
    # tolerance vs float64 by products per tile: fp32 MFMA / 6 products are fp32-accurate,
    # 3 products keep terms down to 2^-8 * 2^-8, 1 product is plain bf16
    tol = {0: 2e-6, 6: 2e-6, 3: 3e-5, 1: 2e-2}[split]
    import ctypes
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    from deformcontact_amd.ops import _i64_array, _ptr_array
    L = _lib.lib()
    slab = torch.from_numpy(hashed_uniform((n, nseg * fi), 5, 2.0)).to(DEV)
    xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    ld = [nseg * fi] * nseg
    ws = [torch.from_numpy(hashed_uniform((fo, fi), 40 + s, 2.0 / np.sqrt(fi))).to(DEV) for s in range(nseg)]
    bias = torch.from_numpy(hashed_uniform((fo,), 77, 0.5)).to(DEV)
    out = torch.empty(n, fo, device=DEV)
    st = current_stream_ptr(torch.device(DEV))
    fargs = (_ptr_array(xs), _i64_array(ld), _ptr_array(ws), nseg, bias.data_ptr(), int(relu),
             out.data_ptr(), fo, n, fi, fo)
    _lib.check(L.dc_tag_linear_fwd_split(*fargs, split, st) if split else L.dc_tag_linear_fwd(*fargs, st),
               "fwd")
    ref = sum(xs[s].double().cpu() @ ws[s].double().cpu().t() for s in range(nseg)) + bias.double().cpu()
    pre = ref.clone()
    if relu:
        ref = ref.clamp_min(0)
    assert rel_err(_np(out), ref.numpy()) < tol
    # backward
    g = torch.from_numpy(hashed_uniform((n, fo), 91, 2.0)).to(DEV)
    gm = g.double().cpu() * ((pre > 0).double() if relu else 1.0)
    # ReLU mask is taken from the forward output (> 0), as torch's threshold_backward does
    if relu:
        gm = g.double().cpu() * (out.double().cpu() > 0).double()
    gws = [torch.empty(fo, fi, device=DEV) for _ in range(nseg)]
    gb = torch.empty(fo, device=DEV)
    nbytes = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    mask = out.data_ptr() if relu else None
    wargs = (g.data_ptr(), fo, mask, fo, _ptr_array(xs), _i64_array(ld), nseg, _ptr_array(gws), nseg,
             fi, gb.data_ptr(), 0, scratch.data_ptr(), nbytes, n, fi, fo)
    _lib.check(L.dc_tag_linear_bwd_dw_split(*wargs, split, st) if split
               else L.dc_tag_linear_bwd_dw(*wargs, st), "dw")
    for s in range(nseg):
        assert rel_err(_np(gws[s]), (gm.t() @ xs[s].double().cpu()).numpy()) < tol, s
    assert rel_err(_np(gb), gm.sum(0).numpy()) < 2e-6          # bias sums stay fp32 VALU
    gslab = torch.zeros(n, nseg * fi, device=DEV)
    gxs = [gslab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    if split:
        wsb = L.dc_tag_linear_bwd_dx_split_workspace_bytes(fi, fo, nseg)
        wsx = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        _lib.check(L.dc_tag_linear_bwd_dx_split(g.data_ptr(), fo, mask, fo, _ptr_array(ws), nseg,
                                                _ptr_array(gxs), _i64_array(ld), wsx.data_ptr(), wsb,
                                                n, fi, fo, split, st), "dx_split")
    else:
        _lib.check(L.dc_tag_linear_bwd_dx(g.data_ptr(), fo, mask, fo, _ptr_array(ws), nseg, _ptr_array(gxs),
                                          _i64_array(ld), n, fi, fo, st), "dx")
    for s in range(nseg):
        assert rel_err(_np(gxs[s]), (gm @ ws[s].double().cpu()).numpy()) < tol, s



@pytest.mark.parametrize("n,fi,fo,nseg,relu", [(208, 256, 256, 4, True), (1008, 256, 256, 1, False),
                                               (144, 32, 48, 2, True), (4112, 64, 256, 3, True)])
@pytest.mark.parametrize("regime", ["unit", "wide_rows", "tiny", "huge"])
def test_dense_block_fp16x2_vs_float64(n, fi, fo, nseg, relu, regime):
    """The scaled fp16x2 dense block (3 MFMA products) is fp32-accurate: same 2e-6 bound vs float64
    as the fp32-MFMA / six-product kernels, also per ROW when rows differ by 2^20 in magnitude,
    and for operands far outside the fp16 range (1e-12 / 1e+12)."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    from deformcontact_amd.ops import _i64_array, _ptr_array
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    slab = torch.from_numpy(hashed_uniform((n, nseg * fi), 5, 2.0)).to(DEV)
    g = torch.from_numpy(hashed_uniform((n, fo), 91, 2.0)).to(DEV)
    wscale = 2.0 / np.sqrt(fi)
    if regime == "wide_rows":
        rs = torch.logspace(-6, 0, n, device=DEV).unsqueeze(1)
        slab, g = slab * rs, g * rs.flip(0)
    elif regime == "tiny":
        slab, g, wscale = slab * 1e-12, g * 1e-12, wscale * 1e-6
    elif regime == "huge":
        slab, g, wscale = slab * 1e12, g * 1e10, wscale * 1e3
    slab, g = slab.contiguous(), g.contiguous()
    xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    ld = [nseg * fi] * nseg
    ws = [torch.from_numpy(hashed_uniform((fo, fi), 40 + s, wscale)).to(DEV) for s in range(nseg)]
    bias = torch.from_numpy(hashed_uniform((fo,), 77, 0.5)).to(DEV) * float(slab.abs().max()) * wscale
    rowmax = slab.abs().amax(1).contiguous()
    ref = sum(xs[s].double().cpu() @ ws[s].double().cpu().t() for s in range(nseg)) + bias.double().cpu()
    if relu:
        ref = ref.clamp_min(0)

    def row_rel(a, b):                       # worst row, each row on its own scale
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        den = np.abs(b).max(axis=1, keepdims=True)
        den[den == 0] = 1.0
        return float((np.abs(a - b) / den).max())
    wmax = ops.weight_rowmax(ws)
    assert torch.equal(wmax, torch.stack([w.abs().amax(1) for w in ws]).amax(0))
    # the one-launch weight preparation: same row maxima + transposed blocks + their row maxima
    wm2, wtm = torch.empty(fo, device=DEV), torch.empty(fi, device=DEV)
    wimg, wtimg = torch.empty(fo, nseg * fi, device=DEV), torch.empty(fi, nseg * fo, device=DEV)
    _lib.check(L.dc_tag_weight_prep(_ptr_array(ws), nseg, fo, fi, wm2.data_ptr(), wimg.data_ptr(),
                                    wtimg.data_ptr(), wtm.data_ptr(), st), "weight_prep")
    assert torch.equal(wm2, wmax)
    assert torch.equal(wtm, torch.stack([w.abs().amax(0) for w in ws]).amax(0))

    def unsplit(img, rows, rowmax):          # {h1[16], h2[16]} records -> (h1 + h2) / 2^e per row
        rec = img.view(torch.float16).view(rows, -1, 2, 16).double()
        e = torch.frexp(rowmax.double())[1].clamp(min=15 - 126)       # rowmax in [2^(e-1), 2^e)
        return (rec[:, :, 0] + rec[:, :, 1]).reshape(rows, -1) / torch.pow(2.0, (15 - e).double()).unsqueeze(1)
    wcat_ref = torch.cat(ws, dim=1).double()
    wtcat_ref = torch.cat([w.t() for w in ws], dim=1).double()
    assert float((unsplit(wimg, fo, wmax) - wcat_ref).abs().max()) <= 3e-7 * float(wcat_ref.abs().max())
    assert float((unsplit(wtimg, fi, wtm) - wtcat_ref).abs().max()) <= 3e-7 * float(wtcat_ref.abs().max())
    # the same forward as ONE segment over the whole slab with the pre-split weights (k_fwd_h2, weights
    # by LDS-DMA) ...
    out1 = torch.empty(n, fo, device=DEV)
    _lib.check(L.dc_tag_linear_fwd_h2p(slab.data_ptr(), nseg * fi, wimg.data_ptr(), bias.data_ptr(), int(relu),
                                       out1.data_ptr(), fo, n, nseg * fi, fo, rowmax.data_ptr(),
                                       wmax.data_ptr(), None, 0, st), "fwd_h2p")
    # cut reduction (forced by a workspace; no bias / relu): sum of range partials, same 2e-6 bound
    wsb = L.dc_tag_linear_fwd_h2p_workspace_bytes(n, nseg * fi, fo)
    if wsb > 0:
        wsk = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        out3 = torch.empty(n, fo, device=DEV)
        _lib.check(L.dc_tag_linear_fwd_h2p(slab.data_ptr(), nseg * fi, wimg.data_ptr(), None, 0,
                                           out3.data_ptr(), fo, n, nseg * fi, fo, rowmax.data_ptr(),
                                           wmax.data_ptr(), wsk.data_ptr(), wsb, st), "fwd_h2p split-K")
        ref3 = sum(xs[s].double().cpu() @ ws[s].double().cpu().t() for s in range(nseg))
        assert row_rel(_np(out3), ref3.numpy()) < 2e-6
    assert row_rel(_np(out1), ref.numpy()) < 2e-6
    # ... and with the fp32 weights concatenated along K through the generic entry (same kernel,
    # splitting the weights itself): the same planes, so bit-identical
    wcat = torch.cat(ws, dim=1).contiguous()
    out2 = torch.empty(n, fo, device=DEV)
    _lib.check(L.dc_tag_linear_fwd_h2(_ptr_array([slab]), _i64_array([nseg * fi]), _ptr_array([wcat]), 1,
                                      bias.data_ptr(), int(relu), out2.data_ptr(), fo, n, nseg * fi, fo,
                                      rowmax.data_ptr(), wmax.data_ptr(), st), "fwd_h2 one segment")
    assert torch.equal(out1, out2)
    out = torch.empty(n, fo, device=DEV)
    _lib.check(L.dc_tag_linear_fwd_h2(_ptr_array(xs), _i64_array(ld), _ptr_array(ws), nseg, bias.data_ptr(),
                                      int(relu), out.data_ptr(), fo, n, fi, fo, rowmax.data_ptr(),
                                      wmax.data_ptr(), st), "fwd_h2")
    assert row_rel(_np(out), ref.numpy()) < 2e-6
    # dX: rows of g scaled by their own maxima inside the kernel (also returned)
    mask = out.data_ptr() if relu else None
    gm = g.double().cpu() * ((out.double().cpu() > 0).double() if relu else 1.0)
    gslab = torch.zeros(n, nseg * fi, device=DEV)
    gxs = [gslab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    wsb = L.dc_tag_linear_bwd_dx_split_workspace_bytes(fi, fo, nseg)
    wsx = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    growmax = ops.rowabsmax(g)                                     # unmasked: an upper bound is enough
    assert torch.equal(growmax, g.abs().amax(1))
    _lib.check(L.dc_tag_linear_bwd_dx_h2(g.data_ptr(), fo, mask, fo, _ptr_array(ws), nseg, _ptr_array(gxs),
                                         _i64_array(ld), wsx.data_ptr(), wsb, n, fi, fo, growmax.data_ptr(),
                                         wmax.data_ptr(), st), "dx_h2")
    for s in range(nseg):
        assert row_rel(_np(gxs[s]), (gm @ ws[s].double().cpu()).numpy()) < 2e-6, s
    # dW: one scale per operand and node chunk (the contraction runs over the nodes)
    gws = [torch.empty(fo, fi, device=DEV) for _ in range(nseg)]
    gb = torch.empty(fo, device=DEV)
    nbytes = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(L.dc_tag_linear_bwd_dw_h2(g.data_ptr(), fo, mask, fo, _ptr_array(xs), _i64_array(ld), nseg,
                                         _ptr_array(gws), nseg, fi, gb.data_ptr(), 0, scratch.data_ptr(),
                                         nbytes, n, fi, fo, growmax.data_ptr(), rowmax.data_ptr(), st), "dw_h2")
    for s in range(nseg):
        assert rel_err(_np(gws[s]), (gm.t() @ xs[s].double().cpu()).numpy()) < 2e-6, s
    assert rel_err(_np(gb), gm.sum(0).numpy()) < 2e-6


def test_hop_rowmax_variant_bit_identical_and_exact_maxima():
    n, e, f = 333, 2500, 256
    ei = random_multigraph(n, e, 21)
    g = GraphIndex(torch.from_numpy(ei).to(DEV), n)
    slab = torch.from_numpy(hashed_uniform((n, 4 * f), 3, 2.0)).to(DEV)
    ref = slab.clone()
    rowmax = torch.empty(n, device=DEV)
    ops.chained_hops(g, slab, f, 3, backward=False, rowmax=rowmax)
    ops.chained_hops(g, ref, f, 3, backward=False)
    assert torch.equal(slab, ref)
    assert torch.equal(rowmax, ref.abs().amax(1))
    # odd width / unaligned view: scalar path
    x = torch.from_numpy(hashed_uniform((n, 37), 4, 2.0)).to(DEV)
    rm = torch.zeros(n, device=DEV)
    y = ops.hop(g.fwd, x, rowmax=rm, rowmax_mode=0)
    assert torch.equal(y, ops.hop(g.fwd, x)) and torch.equal(rm, y.abs().amax(1))


# --------------------------------------------------------------------------- #
# conv layers: forward + backward vs the oracle
# --------------------------------------------------------------------------- #
def _pair(kind, fi, fo, seed):
    cpu = getattr(pyg_ref, kind)(fi, fo)
    fill_state_dict_(cpu, salt0=seed)
    gpu = getattr(dc.nn, kind)(fi, fo)
    gpu.load_state_dict(cpu.state_dict())
    return cpu, gpu.to(DEV)


@pytest.mark.parametrize("kind", ["TAGConv", "GCNConv", "GATConv"])
@pytest.mark.parametrize("n,e,fi,fo", [(120, 900, 21, 256), (120, 900, 25, 32), (90, 700, 256, 256),
                                       (64, 300, 32, 32)])
def test_conv_forward_backward_vs_oracle(kind, n, e, fi, fo):
    torch.set_num_threads(1)
    ei = random_multigraph(n, e, fi + fo)
    x = hashed_uniform((n, fi), 31, 2.0)
    gup = hashed_uniform((n, fo), 37, 2.0)
    cpu, gpu = _pair(kind, fi, fo, fi)
    xc = torch.from_numpy(x).requires_grad_(True)
    oc = cpu(xc, torch.from_numpy(ei))
    (oc * torch.from_numpy(gup)).sum().backward()
    xg = torch.from_numpy(x).to(DEV).requires_grad_(True)
    og = gpu(xg, torch.from_numpy(ei).to(DEV))
    (og * torch.from_numpy(gup).to(DEV)).sum().backward()
    # float64 evaluation of the same op sequence: the truth both fp32 results are judged against
    import copy
    c64 = copy.deepcopy(cpu).double()
    c64.zero_grad()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    o64 = c64(x64, torch.from_numpy(ei))
    (o64 * torch.from_numpy(gup).double()).sum().backward()
    assert rel_err(_np(og), _np(oc)) < TOL
    assert_parity(_np(og), _np(oc), _np(o64), TOL, "per-row forward", metric=row_rel_err)
    assert rel_err(_np(xg.grad), _np(xc.grad)) < TOL
    gc, g64 = dict(cpu.named_parameters()), dict(c64.named_parameters())
    for name, p in gpu.named_parameters():
        assert_parity(_np(p.grad), _np(gc[name].grad), _np(g64[name].grad), TOL, name)


@pytest.mark.parametrize("path", ["fp32_mfma", "bf16x3"])
def test_tagconv_alternative_dense_paths_vs_oracle(path):
    """The dense-block implementations the default (scaled fp16x2) replaces stay selectable
    (DC_DENSE_SPLIT=0: the auditable strict-fp32 line of bench.py; DC_DENSE_F16X2=0): same parity bar on a wide layer."""
    saved = (ops.DENSE_SPLIT_BF16, ops.DENSE_F16X2)
    try:
        ops.DENSE_SPLIT_BF16 = path != "fp32_mfma"
        ops.DENSE_F16X2 = False
        test_conv_forward_backward_vs_oracle("TAGConv", 150, 1100, 256, 256)
    finally:
        ops.DENSE_SPLIT_BF16, ops.DENSE_F16X2 = saved


@pytest.mark.parametrize("backbone,fname", [("TAGConv", "graphnet_tag_h32.npz"),
                                             ("GCNConv", "graphnet_gcn_h32.npz"),
                                             ("GATConv", "graphnet_gat_h32.npz")])
@pytest.mark.parametrize("fused_attn", [False, True])
def test_config0_full_model_golden(backbone, fname, fused_attn):
    """Fixtures produced by the reference's GraphNet + losses (oracle/make_golden.py)."""
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, gradient_consistency_loss, load_model
    z = load_golden(fname)
    m = load_model(dict(EVERYDAY_NETWORK, hidden_dim=int(z["hidden"]), backbone=backbone))
    fill_state_dict_(m)
    m = m.to(DEV).train()
    m.multihead_attention.fused = "1" if fused_attn else "0"
    rest, rig = golden_graphs(z, DEV)
    acts = {}
    for br in ("resting", "rigid"):
        for li, conv in enumerate(getattr(m, f"conv_layers_{br}")):
            conv.register_forward_hook(
                lambda mod, i, o, k=f"conv_{br}_{li}": acts.__setitem__(k, o.detach()))
    pred = m(rest, rig)
    for k, v in acts.items():
        # every backbone runs with the encoder's ReLU fused into its last kernel (TAGConv: the MFMA epilogue; GCNConv /
        # GATConv: the aggregation's epilogue): a layer's hook sees relu(conv(x)), the fixture holds conv(x)
        ref = z[k].clip(min=0)
        assert rel_err(_np(v), ref) < TOL, k
    assert rel_err(_np(pred.pos), z["pred_pos"]) < TOL
    pred.pos = pred.pos - rest.pos
    tgt = rest.clone()
    tgt.pos = torch.from_numpy(z["def_pos"]).to(DEV) - rest.pos
    l1 = torch.nn.functional.l1_loss(pred.pos, tgt.pos)
    gcl = gradient_consistency_loss(pred, tgt)
    assert abs(float(l1) - float(z["loss_l1"])) <= TOL * abs(float(z["loss_l1"]))
    assert abs(float(gcl) - float(z["loss_gcl"])) <= TOL * abs(float(z["loss_gcl"]))
    (l1 + gcl).backward()
    # float64 truth: the same network (oracle convs) evaluated in double on the fixture's inputs
    m64 = load_model(dict(EVERYDAY_NETWORK, hidden_dim=int(z["hidden"]), backbone=backbone),
                     conv_module=pyg_ref)
    fill_state_dict_(m64)
    m64 = m64.double().train()
    r64, g64_ = golden_graphs(z)
    for b_ in (r64, g64_):
        b_.x, b_.pos = b_.x.double(), b_.pos.double()
    p64 = m64(r64, g64_)
    p64.pos = p64.pos - r64.pos
    t64 = r64.clone()
    t64.pos = torch.from_numpy(z["def_pos"]).double() - r64.pos
    (torch.nn.functional.l1_loss(p64.pos, t64.pos) + gradient_consistency_loss(p64, t64)).backward()
    truth = {k: v.grad.numpy() for k, v in m64.named_parameters()}
    for name, p in m.named_parameters():
        ref = z["grad." + name]
        if name.endswith("att_dst"):
            # softmax is invariant to a per-destination shift: d loss / d att_dst only flows
            # through the leaky-relu kink and is ~1e-12 (rounding noise of the softmax
            # backward); compare it on the scale of its sibling att_src gradient instead.
            scale = np.abs(z["grad." + name.replace("att_dst", "att_src")]).max()
            e_h = np.abs(_np(p.grad) - truth[name]).max()
            e_o = np.abs(ref - truth[name]).max()
            record_parity(name + " (abs, on att_src's scale)", None, True, e_h / scale, e_o / scale,
                          special="a gradient that is mathematically zero, compared on its sibling's scale")
            assert e_h <= max(2 * e_o, TOL * scale), (name, e_h, e_o, scale)
            continue
        # gradients that pass through several layers / the softmax backward dS = P * (dP - delta)
        # (which cancels the common part of dP in fp32 in BOTH computations) can differ between two
        # correct fp32 evaluations by more than 1e-5: accepted only where the float64 evaluation
        # shows the HIP result is within 1e-5 of the truth
        d = assert_parity(_np(p.grad), ref, truth[name], TOL, name)
        assert d < 1e-4, (name, d)


def test_encoder_golden_hidden256():
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model
    z = load_golden("encoder_tag_h256.npz")
    m = load_model(EVERYDAY_NETWORK)
    fill_state_dict_(m)
    m = m.to(DEV).train()
    rest, rig = golden_graphs(z, DEV)
    x_rest, x_rig = m.encode(rest, rig)
    assert rel_err(_np(x_rest), z["enc_rest"]) < TOL
    assert rel_err(_np(x_rig), z["enc_rig"]) < TOL
    assert_parity(_np(x_rest), z["enc_rest"], None, TOL, "enc_rest per row", metric=row_rel_err)
    assert_parity(_np(x_rig), z["enc_rig"], None, TOL, "enc_rig per row", metric=row_rel_err)
    g_rest = torch.from_numpy(hashed_uniform(tuple(x_rest.shape), 991, 2.0)).to(DEV)
    g_rig = torch.from_numpy(hashed_uniform(tuple(x_rig.shape), 997, 2.0)).to(DEV)
    ((x_rest * g_rest).sum() + (x_rig * g_rig).sum()).backward()
    # float64 truth of the encoder gradients (oracle convs in double, same weights and inputs)
    from deformcontact_amd.graphnet import ContactEncoder
    e64 = ContactEncoder([21, 25], 256, conv_module=pyg_ref)
    e64.load_state_dict({k: v for k, v in m.state_dict().items() if k.startswith("conv_layers")})
    e64 = e64.double()
    r64, q64 = golden_graphs(z)
    r64.x, q64.x = r64.x.double(), q64.x.double()
    a64, b64 = e64(r64, q64)
    ((a64 * g_rest.cpu().double()).sum() + (b64 * g_rig.cpu().double()).sum()).backward()
    truth = {k: v.grad.numpy() for k, v in e64.named_parameters()}
    for name, p in m.named_parameters():
        if not name.startswith("conv_layers"):
            continue
        g = _np(p.grad)
        if "grad." + name in z:
            assert_parity(g, z["grad." + name], truth[name], TOL, name)
        else:
            assert_parity(g.reshape(-1)[::97], z["gradprobe." + name], truth[name].reshape(-1)[::97], TOL, name)
            ref = float(z["gradsum." + name])
            assert abs(float(g.astype(np.float64).sum()) - ref) < 1e-4 * max(abs(ref), np.abs(g).sum() * 1e-3)


# --------------------------------------------------------------------------- #
# full BASELINE size (B=32): size-independent properties
# --------------------------------------------------------------------------- #
@pytest.fixture(scope="module")
def everyday_b32():
    from deformcontact_amd import synth
    rest, deff, rig = synth.make_batch(32)
    return rest.to(DEV), rig.to(DEV)


def test_config1_b32_csr_and_hop_properties(everyday_b32):
    rest, rig = everyday_b32
    for b, n_exp, e_exp in ((rest, 32768, 196224), (rig, 24384, 145920)):
        ei = _np(b.edge_index)
        n = b.x.shape[0]
        assert (n, ei.shape[1]) == (n_exp, e_exp)
        g = GraphIndex(b.edge_index, n, validate=True)
        # bit-exact indices at full size
        order = np.argsort(ei[1], kind="stable")
        assert np.array_equal(_np(g.fwd.perm), order.astype(np.int32))
        assert np.array_equal(_np(g.fwd.other), ei[0][order].astype(np.int32))
        assert np.array_equal(_np(g.fwd.ptr)[1:], np.cumsum(np.bincount(ei[1], minlength=n)))
        order_t = np.argsort(ei[0], kind="stable")
        assert np.array_equal(_np(g.bwd.perm), order_t.astype(np.int32))
        # hop on ones = weighted in-degree; sortedness of groups; linearity; adjointness
        f = 256
        ones = torch.ones(n, f, device=DEV)
        y1 = ops.hop(g.fwd, ones)
        wsum = np.zeros(n, np.float64)
        np.add.at(wsum, ei[1][order], _np(g.fwd.w).astype(np.float64))
        assert rel_err(_np(y1)[:, 0], wsum) < 1e-6 and torch.equal(y1[:, 0], y1[:, -1])
        a = torch.from_numpy(hashed_uniform((n, f), 1, 2.0)).to(DEV)
        c = torch.from_numpy(hashed_uniform((n, f), 2, 2.0)).to(DEV)
        lhs = ops.hop(g.fwd, 2.0 * a + c)
        rhs = 2.0 * ops.hop(g.fwd, a) + ops.hop(g.fwd, c)
        assert rel_err(_np(lhs), _np(rhs)) < 1e-6
        yt = ops.hop(g.bwd, c)
        l = float((ops.hop(g.fwd, a).double() * c.double()).sum())
        r = float((a.double() * yt.double()).sum())
        assert abs(l - r) <= 1e-6 * max(abs(l), 1.0)
        # bit-exact vs the C oracle on the whole batch (seconds on one core)
        w_edge = np.zeros(ei.shape[1], np.float32)
        w_edge[_np(g.fwd.perm)] = _np(g.fwd.w)
        assert np.array_equal(_np(ops.hop(g.fwd, a)), hop_c.hop(ei, w_edge, _np(a)))


def test_config1_b32_encoder_vs_oracle(everyday_b32):
    """B=32 encoder forward AND backward (every parameter gradient) against the oracle's ATen-op
    sequence on the host; gradients that differ by more than 1e-5 between the two fp32
    evaluations are judged against the oracle run in float64."""
    from deformcontact_amd.graphnet import ContactEncoder
    rest, rig = everyday_b32
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256)
    ref = ContactEncoder([21, 25], 256, conv_module=pyg_ref)
    ref.load_state_dict(enc.state_dict())
    # 15 M pre-activations: a few of them lie within fp32 rounding of zero, where the fp32 and float64
    # evaluations disagree about the ReLU - one flipped mask element moves a bias gradient by 1e-4 of its
    # maximum, whatever the precision of the sums.  The layer biases are therefore raised so that no
    # pre-activation comes near the kink (checked below); masks themselves are covered at sizes where
    # such ties do not occur (test_conv_forward_backward_vs_oracle, the golden fixtures).
    with torch.no_grad():
        for name, p_ in enc.named_parameters():
            if name.endswith(".bias"):
                p_.fill_(6.0)
            elif ".1.lins." in name:
                p_.mul_(0.1)        # layer 2 sees inputs around 6: keep its sums well inside the bias
    ref.load_state_dict(enc.state_dict())
    ref64 = ContactEncoder([21, 25], 256, conv_module=pyg_ref)
    ref64.load_state_dict(enc.state_dict())
    ref64 = ref64.double()
    enc = enc.to(DEV)
    gen = torch.Generator().manual_seed(5)
    ga = torch.randn(rest.x.shape[0], 256, generator=gen)
    gb = torch.randn(rig.x.shape[0], 256, generator=gen)
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [ga.to(DEV), gb.to(DEV)])
    rest_c = G(rest.x.cpu(), rest.edge_index.cpu())
    rig_c = G(rig.x.cpu(), rig.edge_index.cpu())
    ra, rb = ref(rest_c, rig_c)
    torch.autograd.backward([ra, rb], [ga, gb])
    ta, tb = ref64(G(rest_c.x.double(), rest_c.edge_index), G(rig_c.x.double(), rig_c.edge_index))
    torch.autograd.backward([ta, tb], [ga.double(), gb.double()])
    assert float(ta.min()) > 0.05 and float(tb.min()) > 0.05, "a ReLU came near its kink: raise the biases"
    assert rel_err(_np(a), _np(ra)) < TOL and rel_err(_np(b), _np(rb)) < TOL
    assert_parity(_np(a), _np(ra), _np(ta), TOL, "soft per row", metric=row_rel_err)
    assert_parity(_np(b), _np(rb), _np(tb), TOL, "rigid per row", metric=row_rel_err)
    rp, tp = dict(ref.named_parameters()), dict(ref64.named_parameters())
    for name, p in enc.named_parameters():
        # (at B = 32 the fp32 oracle itself is 1e-4 away from float64 on the bias gradients - 32,768-term
        # sequential fp32 sums - so the HIP-vs-oracle distance alone says nothing there)
        assert_parity(_np(p.grad), _np(rp[name].grad), _np(tp[name].grad), TOL, name)


def test_graph_cache_reuse_and_invalidation():
    clear_cache()
    n = 40
    ei = torch.from_numpy(random_multigraph(n, 200, 9)).to(DEV)
    conv = dc.nn.TAGConv(8, 8).to(DEV)
    g1 = conv.graph(ei, n)
    assert conv.graph(ei, n) is g1
    ei[0, 0] = (ei[0, 0] + 1) % n                          # in-place edit bumps the version
    assert conv.graph(ei, n) is not g1


def test_input_hop_slab_cache_hits_and_invalidates():
    """First-layer hop slab (a function of x and edge_index only) is computed once per
    (x, topology): bit-identical output on a hit, recomputed when x or edge_index change, never
    used when x needs a gradient."""
    from deformcontact_amd.graph import graph_index
    clear_cache()
    n, fi, fo = 150, 21, 64
    ei = torch.from_numpy(random_multigraph(n, 1200, 19)).to(DEV)
    x = torch.from_numpy(hashed_uniform((n, fi), 3, 2.0)).to(DEV)
    conv = dc.nn.TAGConv(fi, fo).to(DEV)
    calls = []
    real = ops._build_input_slab
    ops._build_input_slab = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y0 = conv(x, ei).clone()         # (.clone(): a plain conv call is deferred until its first use, deferred.py)
        y1 = conv(x, ei).clone()
        assert len(calls) == 1 and torch.equal(y0, y1)
        old, ops.HOP_CACHE = ops.HOP_CACHE, False
        y_nc = conv(x, ei).clone()
        ops.HOP_CACHE = old
        assert len(calls) == 2 and torch.equal(y_nc, y0)
        x[3, 2] += 1.0                                        # in-place edit bumps the version
        y2 = conv(x, ei).clone()
        assert len(calls) == 3 and not torch.equal(y2, y0)
        ei2 = ei.clone()
        ei2[0, 5] = (ei2[0, 5] + 1) % n
        y3 = conv(x, ei2).clone()
        assert len(calls) == 4
        xg = x.clone().requires_grad_(True)
        conv(xg, ei).sum().backward()                         # needs grad: no cache, grads flow
        conv(xg, ei).clone()
        assert len(calls) == 6 and xg.grad is not None
        # ahead-of-time (what the loader does): the layer then builds nothing
        x4 = x.clone()
        ops.precompute_input_hops(graph_index(ei, n), x4, 3)
        assert len(calls) == 7
        y4 = conv(x4, ei).clone()
        assert len(calls) == 7 and torch.equal(y4, y2)
    finally:
        ops._build_input_slab = real
    ref = pyg_ref.TAGConv(fi, fo)
    ref.load_state_dict(conv.state_dict())
    assert rel_err(_np(y2), ref(x.cpu(), ei.cpu()).detach().numpy()) < TOL
    assert rel_err(_np(y3), ref(x.cpu(), ei2.cpu()).detach().numpy()) < TOL


def test_topology_cache_reuses_adjacency_for_equal_content_in_new_tensors():
    from deformcontact_amd.graph import graph_index
    from deformcontact_amd.loaders import TopologyCache
    clear_cache()
    n = 200
    a = torch.from_numpy(random_multigraph(n, 1500, 23)).to(DEV)
    b = a.clone()                                             # same topology, fresh tensor
    c = torch.from_numpy(random_multigraph(n, 1500, 24)).to(DEV)
    cache = TopologyCache()
    ga, gb, gc = cache.get(a, n), cache.get(b, n), cache.get(c, n)
    assert ga is gb and gc is not ga and (cache.hits, cache.misses) == (1, 2)
    assert graph_index(b, n) is ga                            # registered for the new tensor
    assert cache.get(a, n, self_loops=True) is not ga         # flags are part of the key
    x = torch.randn(n, 8, device=DEV)
    assert torch.equal(ops.hop(graph_index(b, n).fwd, x), ops.hop(GraphIndex(b, n).fwd, x))


def test_reference_style_usage_through_alias():
    """`from torch_geometric.nn import TAGConv` code path (models/model.py:2) on the GPU."""
    import sys
    dc.install_as_torch_geometric()
    try:
        from torch_geometric.data import Batch, Data
        from torch_geometric.nn import TAGConv
        d = [Data(x=torch.randn(5, 4), edge_index=torch.tensor([[0, 1, 2, 3], [1, 2, 3, 4]]),
                  pos=torch.randn(5, 3)) for _ in range(3)]
        b = Batch.from_data_list(d).to(DEV)
        conv = TAGConv(4, 6).to(DEV)
        out = torch.relu(conv(b.x, b.edge_index))
        out.sum().backward()
        assert out.shape == (15, 6) and conv.lins[3].weight.grad is not None
    finally:
        for k in ("torch_geometric", "torch_geometric.nn", "torch_geometric.data"):
            sys.modules.pop(k, None)


@pytest.mark.parametrize("fi,fo", [(21, 256), (256, 256)])
def test_direct_param_grad_accumulation_equals_autograd(fi, fo):
    """Direct mode (opt-in through ``dp.GradBucket(direct=True)``): dW / bias gradients accumulated
    into the bucket's views by the slab-reduce kernel must equal what autograd's AccumulateGrad
    produces; parameters that merely HAVE a dense ``.grad`` (no bucket) keep stock autograd."""
    from deformcontact_amd import dp
    n, e = 150, 1100
    ei = torch.from_numpy(random_multigraph(n, e, 17)).to(DEV)
    x = torch.from_numpy(hashed_uniform((n, fi), 3, 2.0)).to(DEV).requires_grad_(True)
    gup = torch.from_numpy(hashed_uniform((n, fo), 4, 2.0)).to(DEV)
    conv = dc.nn.TAGConv(fi, fo).to(DEV)
    with torch.no_grad():
        conv.bias.copy_(torch.from_numpy(hashed_uniform((fo,), 5, 0.3)))

    def run():
        out = conv(x, ei, relu=True)
        (out * gup).sum().backward()

    run()                                                   # .grad is None -> autograd path
    ref = {k: p.grad.clone() for k, p in conv.named_parameters()}
    gx_ref = x.grad.clone()
    # a plain dense .grad is NOT a sink: hooks fire, autograd accumulates (2x)
    fired = []
    h = conv.lins[0].weight.register_hook(lambda g: fired.append(1))
    x.grad = None
    run()
    h.remove()
    assert fired, "without a direct bucket tensor hooks must fire"
    for k, p in conv.named_parameters():
        assert rel_err(_np(p.grad), 2 * _np(ref[k])) < 1e-6, k
    # opt in
    bucket = dp.GradBucket(conv.parameters(), direct=True)   # zeroed views
    ptrs = {k: p.grad.data_ptr() for k, p in conv.named_parameters()}
    x.grad = None
    run()
    assert bucket._pending, "direct writes must be reported to the bucket"
    bucket.wait_direct_writes()
    for k, p in conv.named_parameters():
        assert p.grad.data_ptr() == ptrs[k], "direct mode must write in place"
        assert rel_err(_np(p.grad), _np(ref[k])) < 1e-6, k
    assert torch.equal(x.grad, gx_ref)
    x.grad = None
    run()                                                   # accumulates: 2x
    bucket.wait_direct_writes()
    for k, p in conv.named_parameters():
        assert rel_err(_np(p.grad), 2 * _np(ref[k])) < 1e-6, k
    # a frozen parameter takes the layer out of direct mode (nothing may be accumulated into it)
    conv.lins[2].weight.requires_grad_(False)
    frozen = conv.lins[2].weight.grad.clone()
    x.grad = None
    run()
    bucket.wait_direct_writes()
    assert torch.equal(conv.lins[2].weight.grad, frozen)
    assert rel_err(_np(conv.lins[1].weight.grad), 3 * _np(ref["lins.1.weight"])) < 1e-6
    conv.lins[2].weight.requires_grad_(True)
    old = ops.DIRECT_PARAM_GRAD
    try:
        ops.DIRECT_PARAM_GRAD = False
        x.grad = None
        run()                                               # stock autograd accumulate
        assert rel_err(_np(conv.lins[1].weight.grad), 4 * _np(ref["lins.1.weight"])) < 1e-6
    finally:
        ops.DIRECT_PARAM_GRAD = old


def test_direct_param_grad_two_stream_encoder_with_bucket():
    """Direct gradient writes of the rigid branch land on the encoder's side stream: the bucket's
    consumers must be ordered behind them.  Two-stream encoder + GradBucket(direct=True) + FlatAdam
    against the same step with stock autograd accumulation, several steps."""
    from deformcontact_amd import dp, synth
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(3, soft_vertices=200, sphere_resolution=7))
    g_s = torch.randn(rest.x.shape[0], 256, device=DEV)
    g_r = torch.randn(rig.x.shape[0], 256, device=DEV)
    finals = []
    for direct in (False, True):
        torch.manual_seed(0)
        enc = ContactEncoder([21, 25], 256).to(DEV)
        enc.overlap_branches = True
        bucket = dp.GradBucket(enc.parameters(), direct=direct)
        opt = dp.FlatAdam(bucket, lr=1e-3, zero_grad_in_step=True)
        for _ in range(4):
            a, b = enc(rest, rig)
            torch.autograd.backward([a, b], [g_s, g_r])
            bucket.all_reduce_mean()
            opt.step()
        torch.cuda.synchronize()
        assert not bucket._pending
        finals.append(opt.flat_param.clone())
    assert rel_err(_np(finals[1]), _np(finals[0])) < 1e-6


def test_foreign_column_view_is_not_taken_for_a_hop_slab():
    """A caller's own ``feat[:, :F]`` view of a wider buffer has the shape and strides of column
    block 0 of a hop slab; the layer must not run its hops in place there."""
    n, fi, fo = 96, 32, 16
    ei = torch.from_numpy(random_multigraph(n, 500, 5)).to(DEV)
    conv = dc.nn.TAGConv(fi, fo).to(DEV)
    wide = torch.from_numpy(hashed_uniform((n, 4 * fi), 9, 1.0)).to(DEV)
    keep = wide.clone()
    out_view = conv(wide[:, :fi], ei)
    assert torch.equal(wide, keep), "the caller's buffer was overwritten"
    out_copy = conv(keep[:, :fi].contiguous(), ei)
    assert torch.equal(out_view, out_copy)


@pytest.mark.parametrize("fused_zero", [False, True])
def test_flat_adam_matches_torch_adam(fused_zero):
    """dc_adam_flat (one kernel over the flat bucket) vs torch.optim.Adam(lr=4e-4) defaults; with
    ``zero_grad_in_step`` the kernel also clears the gradients (no separate zero_grad)."""
    from deformcontact_amd import dp
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 190), torch.nn.Linear(190, 5)).to(DEV)
    mine = torch.nn.Sequential(torch.nn.Linear(37, 190), torch.nn.Linear(190, 5)).to(DEV)
    mine.load_state_dict(ref.state_dict())
    o_ref = torch.optim.Adam(ref.parameters(), lr=4e-4)
    bucket = dp.GradBucket(mine.parameters())
    o_mine = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=fused_zero)
    x = torch.randn(64, 37, device=DEV)
    for step in range(25):
        o_ref.zero_grad()
        ref(x).square().mean().backward()
        o_ref.step()
        if not fused_zero:
            o_mine.zero_grad()
        mine(x).square().mean().backward()
        bucket.all_reduce_mean()
        o_mine.step()
        if fused_zero:
            assert not bucket.flat.any()
    for a, b in zip(ref.parameters(), mine.parameters()):
        assert rel_err(_np(b), _np(a)) < 1e-6
    assert float(o_mine.step_count[0]) == 25.0 and not o_mine.step_count[1:].any()


@pytest.mark.parametrize("hidden", [21, 32, 40, 256])
def test_encoder_slab_handoff_any_hidden_width(hidden):
    """Layer output written into the next layer's (possibly K-padded) hop slab: vs oracle."""
    from deformcontact_amd import synth
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = synth.make_batch(2, soft_vertices=64, sphere_resolution=4)
    torch.manual_seed(hidden)
    enc = ContactEncoder([21, 25], hidden, encoder_layers=3)
    ref = ContactEncoder([21, 25], hidden, encoder_layers=3, conv_module=pyg_ref)
    ref.load_state_dict(enc.state_dict())
    enc = enc.to(DEV)
    a, b = enc(rest.clone().to(DEV), rig.clone().to(DEV))
    ra, rb = ref(rest, rig)
    assert rel_err(_np(a), _np(ra)) < TOL and rel_err(_np(b), _np(rb)) < TOL
    (a.square().sum() + b.square().sum()).backward()
    (ra.square().sum() + rb.square().sum()).backward()
    ref64 = ContactEncoder([21, 25], hidden, encoder_layers=3, conv_module=pyg_ref)
    ref64.load_state_dict(ref.state_dict())
    ref64 = ref64.double()
    ta, tb = ref64(G(rest.x.double(), rest.edge_index), G(rig.x.double(), rig.edge_index))
    (ta.square().sum() + tb.square().sum()).backward()
    rp, tp = dict(ref.named_parameters()), dict(ref64.named_parameters())
    for name, p in enc.named_parameters():
        assert_parity(_np(p.grad), _np(rp[name].grad), _np(tp[name].grad), TOL, name)


# --------------------------------------------------------------------------- #
# blocked cross-attention vs the reference's formula in float64
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("ns,nr,d,dv,block", [(300, 130, 256, 256, 128), (64, 48, 32, 64, 2048),
                                              (1000, 777, 256, 256, 512), (17, 5, 16, 16, 16),
                                              # >= 1024 keys: padded to a multiple of 128, dK / dV on the 128 x 256
                                              # dW tiles, the two long reductions split over ranges on the wide tiles
                                              (4096, 3048, 256, 256, 2048)])
def test_blocked_attention_core_vs_float64(ns, nr, d, dv, block):
    from deformcontact_amd.attention import attention_core
    torch.manual_seed(ns + nr)
    q = (torch.randn(ns, d, device=DEV) * 0.5).requires_grad_()
    k = (torch.randn(nr, d, device=DEV) * 0.5).requires_grad_()
    v = torch.randn(nr, dv, device=DEV).requires_grad_()
    go = torch.randn(ns, dv, device=DEV)
    out = attention_core(q, k, v, block_rows=block)
    out.backward(go)
    qd, kd, vd = (t.detach().double().cpu().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(qd @ kd.t(), dim=-1) @ vd            # models/model.py:16-18
    ref.backward(go.double().cpu())
    assert rel_err(_np(out), ref.detach().numpy()) < 5e-6
    for name, got, want in (("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        assert rel_err(_np(got), want.numpy()) < 1e-5, name


def test_prefetch_loader_uploads_equal_plain_batches():
    from deformcontact_amd.loaders import (PrefetchLoader, SyntheticEverydayDataset, iterate_batches,
                                           to_batches)
    ds = SyntheticEverydayDataset(5, first_idx=1, soft_vertices=128, sphere_resolution=6)
    plain = [to_batches(c, DEV) for c in iterate_batches(ds, 2)]
    got = [b for _, b in PrefetchLoader(ds, 2, DEV)]
    assert len(got) == len(plain)
    for b0, b1 in zip(plain, got):
        for x, y in zip(b0, b1):
            assert y.x.is_cuda and torch.equal(x.x, y.x) and torch.equal(x.edge_index, y.edge_index)
            assert torch.equal(x.pos, y.pos)


# --------------------------------------------------------------------------- #
# bf16-stored features (BASELINE.json configs[4])
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,k,fo,out_bf16,relu,bias", [(300, 32, 128, False, False, True), (129, 96, 192, False, True, True),
                                                       (1000, 1024, 256, False, True, False), (257, 64, 40, True, True, True),
                                                       (5, 128, 256, True, False, True)])
def test_dense_block_bf16_operands_vs_float64(n, k, fo, out_bf16, relu, bias):
    """dc_tag_linear_fwd_bf16 through the C ABI: bf16 operands, fp32 accumulate.  Against float64 on
    the SAME bf16-rounded operands the only error is fp32 accumulation (<= 1e-5 per row, measured
    ~1e-6); a bf16 output adds one rounding (2^-9).  Ragged N / Fo (tile edges), 1 .. 32 stages,
    A passed as a column window of a wider buffer (lda > K)."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    wide = torch.from_numpy(hashed_uniform((n, k + 64), 11, 2.0)).to(DEV).bfloat16()
    a = wide[:, 32:32 + k]                                   # lda = k + 64, 16-byte aligned window
    w32 = torch.from_numpy(hashed_uniform((fo, k), 12, 0.5)).to(DEV)
    b = torch.from_numpy(hashed_uniform((fo,), 13, 1.0)).to(DEV) if bias else None
    w = torch.empty(fo, k, dtype=torch.bfloat16, device=DEV)
    halves = [w32[:, :k // 2].contiguous(), w32[:, k // 2:].contiguous()]
    rc = L.dc_to_bf16(ops._ptr_array(halves), 2, fo, k // 2, k // 2, w.data_ptr(), k, st)
    assert rc == 0
    assert torch.equal(w, w32.bfloat16())                    # concatenation + round-to-nearest-even
    out = torch.full((n, fo + 3), 7.0, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
    rc = L.dc_tag_linear_fwd_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), b.data_ptr() if bias else None,
                                  int(relu), out.data_ptr(), out.stride(0), int(out_bf16), n, k, fo, st)
    assert rc == 0, L.dc_last_error()
    ref = a.double().cpu() @ w.double().cpu().t()
    if bias:
        ref = ref + b.double().cpu()
    if relu:
        ref = ref.clamp_min(0)
    got = out[:, :fo].double().cpu()
    scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
    err = ((got - ref).abs() / scale).max().item()
    assert err < (6e-3 if out_bf16 else 1e-5), err
    assert (out[:, fo:] == 7.0).all()                        # nothing written past Fo


def _radius100k():
    from deformcontact_amd import synth
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    return pos.to(DEV), ei.to(DEV)


def test_config4_radius100k_bf16_tagconv_forward_vs_float64():
    """configs[4]: one TAGConv(256, 256) layer on the 100k-point radius graph with the features
    stored as bf16 (dc_spmm_bf16 hops, bf16 MFMA dense block), against the float64 closed form
    sum_k A^k X W_k^T + b on the same bf16-rounded inputs.  Stated bf16 tolerance: 2e-2 of max |ref|
    (three chained bf16-stored hops + a bf16 output: a few 2^-9), rms 4e-3.  Also: Morton node
    reordering leaves the result bit-identical."""
    import scipy.sparse as sp
    from deformcontact_amd.graph import NodeOrder
    pos, ei = _radius100k()
    n, f = pos.shape[0], 256
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, f, generator=gen).to(DEV).bfloat16()
    torch.manual_seed(5)
    conv = dc.nn.TAGConv(f, f).to(DEV)
    with torch.no_grad():
        conv.bias.copy_(torch.from_numpy(hashed_uniform((f,), 5, 0.3)))
        y = conv(x, ei, relu=False)
        order = NodeOrder.morton(pos)
        y_m = order.undo(conv(order.apply(x), order.relabel(ei), relu=False))
    assert y.dtype == torch.bfloat16
    assert torch.equal(y, y_m), "node reordering must not change a single bit"
    ei_c = ei.cpu().numpy()
    deg = np.bincount(ei_c[1], minlength=n).astype(np.float64)
    dis = np.zeros_like(deg)
    dis[deg > 0] = deg[deg > 0] ** -0.5
    a = sp.csr_matrix((dis[ei_c[0]] * dis[ei_c[1]], (ei_c[1], ei_c[0])), shape=(n, n))
    xk = x.double().cpu().numpy()
    ref = np.zeros((n, f))
    for k, lin in enumerate(conv.lins):
        if k:
            xk = a @ xk
        ref += xk @ lin.weight.detach().bfloat16().double().cpu().numpy().T
    ref += conv.bias.detach().double().cpu().numpy()
    got = y.double().cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-2
    assert np.sqrt(np.mean((got - ref) ** 2)) / scale < 4e-3


def test_two_layer_bf16_stack_slab_handoff_and_fp32_head():
    """Layer 1 writes its bf16 output straight into layer 2's slab; the last layer can emit fp32."""
    n, f = 700, 64
    ei = torch.from_numpy(random_multigraph(n, 5000, 77)).to(DEV)
    x = torch.from_numpy(hashed_uniform((n, f), 3, 2.0)).to(DEV).bfloat16()
    torch.manual_seed(1)
    c1, c2 = dc.nn.TAGConv(f, f).to(DEV), dc.nn.TAGConv(f, f).to(DEV)
    c2.bf16_out = torch.float32
    with torch.no_grad():
        h = c1(x, ei, relu=True, next_conv=c2)
        assert h._base is not None and h._base.shape[0] == n and h._base.shape[1] >= 4 * f
        y = c2(h, ei, relu=True)
        h_plain = c1(x, ei, relu=True)
        y_plain = c2(h_plain, ei, relu=True)
    assert y.dtype == torch.float32 and torch.equal(h, h_plain) and torch.equal(y, y_plain)
    r1, r2 = pyg_ref.TAGConv(f, f), pyg_ref.TAGConv(f, f)
    r1.load_state_dict(c1.state_dict())
    r2.load_state_dict(c2.state_dict())
    with torch.no_grad():
        ref = torch.relu(r2.double()(torch.relu(r1.double()(x.double().cpu(), ei.cpu())), ei.cpu()))
    assert rel_err(_np(y), ref.numpy()) < 3e-2
@pytest.mark.gpu
@pytest.mark.parametrize("n,fi,fo,nseg,mask", [(3048, 256, 256, 4, False), (3048, 96, 256, 1, True),
                                               (1003, 64, 48, 2, True), (31, 32, 32, 1, False), (7, 32, 32, 1, True)])
def test_dw_ragged_node_count(n, fi, fo, nseg, mask):
    """dW over a node count that is not a multiple of the 16-row stage (the shipped batch of 4 rigid spheres has
    3,048 rows): the whole stages run on the split / fp16x2 kernels, the trailing rows through the generic kernel
    into a partial slot of their own - same result, same bound vs float64 as the aligned case."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    from deformcontact_amd.ops import _i64_array, _ptr_array
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    slab = torch.from_numpy(hashed_uniform((n, nseg * fi), 11, 2.0)).to(DEV)
    g = torch.from_numpy(hashed_uniform((n, fo), 12, 2.0)).to(DEV)
    out = torch.from_numpy(hashed_uniform((n, fo), 13, 1.0)).to(DEV)            # relu mask source
    xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    gm = g.double().cpu() * ((out.double().cpu() > 0).double() if mask else 1.0)
    refs = [gm.t() @ x.double().cpu() for x in xs]
    refb = gm.sum(0)
    nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    mptr = out.data_ptr() if mask else None

    def check(name, launch):
        gws = [torch.full((fo, fi), float("nan"), device=DEV) for _ in range(nseg)]
        gb = torch.full((fo,), float("nan"), device=DEV)
        args = (g.data_ptr(), fo, mptr, fo, _ptr_array(xs), _i64_array([nseg * fi] * nseg), nseg, _ptr_array(gws), nseg,
                fi, gb.data_ptr(), 0, scratch.data_ptr(), nb, n, fi, fo)
        _lib.check(launch(args), name)
        for s in range(nseg):
            assert rel_err(_np(gws[s]), refs[s].numpy()) < 2e-6, (name, s)
        assert rel_err(_np(gb), refb.numpy()) < 2e-6, name
    check("split6", lambda a: L.dc_tag_linear_bwd_dw_split(*a, 6, st))
    check("fp32", lambda a: L.dc_tag_linear_bwd_dw(*a, st))
    if not mask:                                         # the fp16x2 form takes a pre-masked gradient
        gmax, xmax = ops.rowabsmax(g), slab.abs().amax(1).contiguous()
        check("h2", lambda a: L.dc_tag_linear_bwd_dw_h2(*a, gmax.data_ptr(), xmax.data_ptr(), st))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,d,nr", [(2048, 256, 3048), (200, 64, 333), (96, 32, 40)])
def test_scores_exp_epilogue_bit_identical_to_gemm_plus_exp_rows(rows, d, nr):
    """`dc_tag_linear_fwd_h2p_exp` (the attention backward's recompute: scores and exp(s - lse) in one launch) against
    `dc_tag_linear_fwd_h2p` + `dc_attn_exp_rows`: bit-identical, padded key columns exactly zero - on the 128 x 256
    tiles (first shape) and on the 64/128 x 128 kernel (the small ones)."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    from deformcontact_amd.ops import _ptr_array
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    nrp = (nr + 15) // 16 * 16
    q = torch.from_numpy(hashed_uniform((rows, d), 3, 1.0)).to(DEV)
    k = torch.zeros(nrp, d, device=DEV)
    k[:nr] = torch.from_numpy(hashed_uniform((nr, d), 4, 1.0)).to(DEV)
    qmax = ops.rowabsmax(q)
    kmax = torch.empty(nrp, device=DEV)
    kimg = torch.empty(nrp, d, device=DEV)
    _lib.check(L.dc_tag_weight_prep(_ptr_array([k]), 1, nrp, d, kmax.data_ptr(), kimg.data_ptr(), None, None, st), "prep")
    s = torch.empty(rows, nrp, device=DEV)
    _lib.check(L.dc_tag_linear_fwd_h2p(q.data_ptr(), d, kimg.data_ptr(), None, 0, s.data_ptr(), nrp, rows, d, nrp,
                                       qmax.data_ptr(), kmax.data_ptr(), None, 0, st), "scores")
    lse = torch.logsumexp(s[:, :nr].double(), dim=1).float().contiguous()
    ref = s.clone()
    _lib.check(L.dc_attn_exp_rows(ref.data_ptr(), nrp, rows, nr, nrp, lse.data_ptr(), st), "exp_rows")
    out = torch.full((rows, nrp), float("nan"), device=DEV)
    _lib.check(L.dc_tag_linear_fwd_h2p_exp(q.data_ptr(), d, kimg.data_ptr(), out.data_ptr(), nrp, rows, d, nrp,
                                           qmax.data_ptr(), kmax.data_ptr(), lse.data_ptr(), nr, st), "scores_exp")
    assert torch.equal(out, ref)
    assert not out[:, nr:].any()
    assert float((out[:, :nr].double().sum(1) - 1).abs().max()) < 1e-5


@pytest.mark.gpu
def test_flat_adam_keeps_every_parameter_16_byte_aligned():
    """`dp.FlatAdam` re-points the parameters into one flat buffer: with the reference network (whose decoder
    ends in a 3-element bias, defined BEFORE the attention heads) every parameter and gradient view must still
    start on a 16-byte boundary, the padding must stay zero through Adam steps, and the result must equal
    torch.optim.Adam's."""
    from deformcontact_amd import dp
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model
    torch.manual_seed(0)
    model = load_model(EVERYDAY_NETWORK).to(DEV)
    ref = load_model(EVERYDAY_NETWORK).to(DEV)
    ref.load_state_dict(model.state_dict())
    ref_opt = torch.optim.Adam(ref.parameters(), lr=4e-4)
    bucket = dp.GradBucket(model.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=4e-4)
    assert any(p.numel() % 4 for p in bucket.params)             # the network does have an odd-sized parameter
    for p in bucket.params:
        assert p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0
    pad = torch.ones(bucket.numel, dtype=torch.bool, device=DEV)
    for p, off in zip(bucket.params, bucket.offsets):
        pad[off:off + p.numel()] = False
    assert int(pad.sum()) == bucket.numel - sum(p.numel() for p in bucket.params) > 0
    gen = torch.Generator(device="cpu").manual_seed(5)
    for _ in range(3):
        for p, q in zip(bucket.params, [q for q in ref.parameters() if q.requires_grad]):
            g = torch.randn(p.shape, generator=gen).to(DEV)
            p.grad.copy_(g)
            q.grad = g.clone()
        opt.step()
        ref_opt.step()
        assert not opt.flat_param[pad].any() and not opt.exp_avg[pad].any() and not bucket.flat[pad].any()
    for p, q in zip(bucket.params, [q for q in ref.parameters() if q.requires_grad]):
        assert rel_err(_np(p), _np(q)) < TOL          # 1e-5: three fp32 Adam steps, different op order


