"""Both encoder branches as ONE block-diagonal problem: the merged adjacency
(``dc_graph_build_parts``), row windows, the grouped TAGConv layer (``dc_tag_grouped_*``) and the merged
encoder path against the per-branch path - BIT-IDENTICAL wherever the node counts are whole stages, and
against the CPU oracle / float64 through the usual parity policy elsewhere.  Reference call sites:
/root/reference/models/model.py:69-78 (the two encoder loops)."""
import numpy as np
import pytest
import torch

import deformcontact_amd as dc
from deformcontact_amd import ops, synth
from deformcontact_amd.graph import GROUP_ALIGN, GraphIndex, clear_cache, merged_graph_index
from deformcontact_amd.graphnet import ContactEncoder
from oracle import hop_c, pyg_ref
from tests.helpers import assert_parity, random_multigraph, rel_err, row_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _parts(shapes, seed=0):
    return [(torch.from_numpy(random_multigraph(n, e, seed + 17 * i)).to(DEV), n) for i, (n, e) in enumerate(shapes)]


# --------------------------------------------------------------------------- #
# merged adjacency: every part's rows exactly as its own GraphIndex holds them
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("shapes", [[(300, 2000), (77, 500)], [(256, 1000), (512, 3000)], [(1, 0), (5, 9)],
                                    [(1000, 20000), (999, 100), (3, 3)], [(4099, 30011), (70000, 300000)],
                                    [(40, 100)], [(600, 9000), (500, 70000), (700, 100), (256, 256)]])
def test_merged_adjacency_equals_per_part_adjacencies(shapes):
    parts = _parts(shapes)
    mg = GraphIndex.from_parts(parts)
    mg.validate()
    assert mg.num_nodes % GROUP_ALIGN == 0 and len(mg.row_beg) == len(parts)
    ptr_f, ptr_b = _np(mg.fwd.ptr), _np(mg.bwd.ptr)
    for gi, (ei, n) in enumerate(parts):
        g = GraphIndex(ei, n)
        r0, e0, e = mg.row_beg[gi], mg.edge_beg[gi], ei.shape[1]
        assert r0 % GROUP_ALIGN == 0
        for adj, madj, mptr in ((g.fwd, mg.fwd, ptr_f), (g.bwd, mg.bwd, ptr_b)):
            p0 = int(mptr[r0])
            assert np.array_equal(mptr[r0:r0 + n + 1] - p0, _np(adj.ptr))
            assert np.array_equal(_np(madj.other)[p0:p0 + e] - r0, _np(adj.other)[:e])
            assert np.array_equal(_np(madj.perm)[p0:p0 + e] - e0, _np(adj.perm)[:e])
            assert np.array_equal(_np(madj.w)[p0:p0 + e].view(np.int32), _np(adj.w)[:e].view(np.int32))
        # and the oracle's scalar C build of the part
        ptr, other, perm = hop_c.csr_build(_np(ei), n, 1)
        p0 = int(ptr_f[r0])
        assert np.array_equal(ptr_f[r0:r0 + n + 1] - p0, ptr)
        assert np.array_equal(_np(mg.fwd.other)[p0:p0 + e] - r0, other)
        # padding rows behind the part are isolated
        r1 = mg.row_beg[gi + 1] if gi + 1 < len(parts) else mg.num_nodes
        assert np.all(np.diff(ptr_f[r0 + n:r1 + 1]) == 0) and np.all(np.diff(ptr_b[r0 + n:r1 + 1]) == 0)
    assert int(ptr_f[-1]) == sum(ei.shape[1] for ei, _ in parts)


def test_merged_adjacency_flags_out_of_range_ids_per_part():
    a = torch.from_numpy(random_multigraph(100, 500, 1)).to(DEV)
    b = torch.from_numpy(random_multigraph(50, 300, 2)).to(DEV)
    b[0, 7] = 50                           # outside part 1's own range although inside the merged node space
    mg = GraphIndex.from_parts([(a, 100), (b, 50)])
    with pytest.raises(IndexError):
        mg.validate()


@pytest.mark.parametrize("f", [21, 25, 32, 100, 256])
def test_window_hop_equals_standalone_hop_bitwise(f):
    parts = _parts([(700, 5000), (333, 2500), (256, 999)], seed=5)
    mg = merged_graph_index(parts)
    for gi, (ei, n) in enumerate(parts):
        g = GraphIndex(ei, n)
        w = mg.window(gi)
        assert w.fwd.row_offset == mg.row_beg[gi]
        x = torch.randn(n, f, device=DEV)
        add = torch.randn(n, f, device=DEV)
        for adj_w, adj_g in ((w.fwd, g.fwd), (w.bwd, g.bwd)):
            assert torch.equal(ops.hop(adj_w, x), ops.hop(adj_g, x))
            assert torch.equal(ops.hop(adj_w, x, addend=add), ops.hop(adj_g, x, addend=add))
            for mode in (0, 1, 3):                   # ... and with the row maxima the wide dense block scales by
                rm_w, rm_g = torch.full((n,), 0.5, device=DEV), torch.full((n,), 0.5, device=DEV)
                assert torch.equal(ops.hop(adj_w, x, rowmax=rm_w, rowmax_mode=mode),
                                   ops.hop(adj_g, x, rowmax=rm_g, rowmax_mode=mode))
                assert torch.equal(rm_w, rm_g)
        # oracle: the scalar C hop over the part's own edge list
        e = ei.shape[1]
        w_edge = np.zeros(e, np.float32)
        w_edge[_np(g.fwd.perm)[:e]] = _np(g.fwd.w)[:e]
        ref = hop_c.hop(_np(ei), w_edge, _np(x))
        assert np.array_equal(_np(ops.hop(w.fwd, x)), ref)


def test_merged_hop_equals_per_part_hops_bitwise_with_row_maxima():
    parts = _parts([(1024, 6132), (762, 4560)], seed=9)
    mg = GraphIndex.from_parts(parts)
    f = 256
    x = torch.zeros(mg.num_nodes, f, device=DEV)
    xs = []
    for gi, (_, n) in enumerate(parts):
        xs.append(torch.randn(n, f, device=DEV))
        x[mg.row_beg[gi]:mg.row_beg[gi] + n] = xs[-1]
    rm = torch.empty(mg.num_nodes, device=DEV)
    y = ops.hop(mg.fwd, x, rowmax=rm, rowmax_mode=1)
    for gi, (ei, n) in enumerate(parts):
        g = GraphIndex(ei, n)
        rm_g = torch.empty(n, device=DEV)
        y_g = ops.hop(g.fwd, xs[gi], rowmax=rm_g, rowmax_mode=1)
        r0 = mg.row_beg[gi]
        assert torch.equal(y[r0:r0 + n], y_g) and torch.equal(rm[r0:r0 + n], rm_g)
        r1 = mg.row_beg[gi + 1] if gi + 1 < len(parts) else mg.num_nodes
        assert not y[r0 + n:r1].any() and not rm[r0 + n:r1].any()           # padding rows stay zero


# --------------------------------------------------------------------------- #
# grouped TAGConv layer vs one launch set per branch
# --------------------------------------------------------------------------- #
def _layer_pair(seed, fi=256, fo=256):
    torch.manual_seed(seed)
    a, b = dc.nn.TAGConv(fi, fo).to(DEV), dc.nn.TAGConv(fi, fo).to(DEV)
    with torch.no_grad():
        for c in (a, b):
            c.bias.uniform_(-0.3, 0.3)
    return a, b


def _run_separate(convs, parts, xs, gouts):
    outs, gxs, grads = [], [], []
    for c, (ei, n), x, go in zip(convs, parts, xs, gouts):
        c.zero_grad(set_to_none=True)
        x = x.clone().requires_grad_(True)
        o = c(x, ei, relu=True)
        o.backward(go)
        outs.append(o.detach())
        gxs.append(x.grad.clone())
        grads.append([p.grad.clone() for p in c.parameters()])
    return outs, gxs, grads


def _run_grouped(convs, parts, xs, gouts):
    mg = merged_graph_index(parts)
    for c in convs:
        c.zero_grad(set_to_none=True)
    xin = [x.clone().requires_grad_(True) for x in xs]
    outs = ops.tag_conv_grouped(mg, xin, [[lin.weight for lin in c.lins] for c in convs],
                                [c.bias for c in convs], relu=True)
    torch.autograd.backward(list(outs), list(gouts))
    return ([o.detach() for o in outs], [x.grad.clone() for x in xin],
            [[p.grad.clone() for p in c.parameters()] for c in convs])


@pytest.mark.parametrize("shapes", [[(1024, 6132), (768, 4600)], [(512, 3000), (256, 1500), (1280, 9000)],
                                    [(2048, 12264), (1536, 9120)]])
def test_grouped_layer_bit_identical_to_per_branch_layers(shapes):
    """Node counts that are whole 32-row stages: outputs, input gradients AND parameter gradients are the
    same bits as the per-branch launches (same tiles, same reduction order, same node chunks)."""
    clear_cache()
    parts = _parts(shapes, seed=3)
    convs = [_layer_pair(10 + i)[0] for i in range(len(parts))]
    xs = [torch.randn(n, 256, device=DEV).relu_() for _, n in parts]
    gouts = [torch.randn(n, 256, device=DEV) for _, n in parts]
    o_s, gx_s, gr_s = _run_separate(convs, parts, xs, gouts)
    o_g, gx_g, gr_g = _run_grouped(convs, parts, xs, gouts)
    for i in range(len(parts)):
        assert torch.equal(o_s[i], o_g[i]), f"output of group {i}"
        assert torch.equal(gx_s[i], gx_g[i]), f"input gradient of group {i}"
        for a, b, (name, _) in zip(gr_s[i], gr_g[i], convs[i].named_parameters()):
            assert torch.equal(a, b), f"gradient of {name} of group {i}"


def test_grouped_layer_ragged_node_counts_vs_float64():
    """Node counts that are NOT whole stages (the shipped batch of 4 spheres: 3,048 rows): the padding rows
    take the place of the per-branch path's generic tail kernel - same maths, different summation order -
    so this checks against float64 under the parity policy, per row for outputs / input gradients."""
    clear_cache()
    shapes = [(1000, 6000), (3048, 18240)]
    parts = _parts(shapes, seed=4)
    convs = list(_layer_pair(20))
    xs = [torch.randn(n, 256, device=DEV).relu_() for _, n in parts]
    gouts = [torch.randn(n, 256, device=DEV) for _, n in parts]
    o_g, gx_g, gr_g = _run_grouped(convs, parts, xs, gouts)
    for i, (c, (ei, n)) in enumerate(zip(convs, parts)):
        ref = pyg_ref.TAGConv(256, 256)
        ref.load_state_dict({k: v.cpu() for k, v in c.state_dict().items()})
        ref64 = pyg_ref.TAGConv(256, 256).double()
        ref64.load_state_dict({k: v.cpu().double() for k, v in c.state_dict().items()})
        res = []
        for m, dt in ((ref, torch.float32), (ref64, torch.float64)):
            x = xs[i].cpu().to(dt).requires_grad_(True)
            o = torch.relu(m(x, ei.cpu()))
            o.backward(gouts[i].cpu().to(dt))
            res.append((o.detach(), x.grad, [p.grad for p in m.parameters()]))
        (o32, gx32, gp32), (o64, gx64, gp64) = res
        assert_parity(_np(o_g[i]), _np(o32), _np(o64), name=f"out[{i}]", metric=row_rel_err)
        assert_parity(_np(gx_g[i]), _np(gx32), _np(gx64), name=f"gx[{i}]")
        for a, b, t, (name, _) in zip(gr_g[i], gp32, gp64, c.named_parameters()):
            assert_parity(_np(a), _np(b), _np(t), name=f"{name}[{i}]")


def test_grouped_layer_accepts_inputs_that_are_not_slab_views_and_partial_grads():
    clear_cache()
    parts = _parts([(512, 3000), (256, 1500)], seed=6)
    convs = list(_layer_pair(30))
    xs = [torch.randn(n, 256, device=DEV) for _, n in parts]
    mg = merged_graph_index(parts)
    # no input gradient requested, only one branch's output used downstream
    outs = ops.tag_conv_grouped(mg, xs, [[lin.weight for lin in c.lins] for c in convs], [c.bias for c in convs],
                                relu=False)
    outs[1].sum().backward()
    assert convs[0].bias.grad is None or not convs[0].bias.grad.any()
    ref = dc.nn.TAGConv(256, 256).to(DEV)
    ref.load_state_dict(convs[1].state_dict())
    o = ref(xs[1], parts[1][0])
    o.sum().backward()
    assert torch.equal(o, outs[1])
    for a, b in zip(convs[1].parameters(), ref.parameters()):
        assert torch.equal(a.grad, b.grad)


# --------------------------------------------------------------------------- #
# the encoder: merged path vs per-branch path
# --------------------------------------------------------------------------- #
def _encoder_run(enc, rest, rig, g_rest, g_rig, merged):
    enc.merge_branches = merged
    clear_cache()
    enc.zero_grad(set_to_none=True)
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [g_rest, g_rig])
    torch.cuda.synchronize()
    return a.detach().clone(), b.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}


@pytest.mark.parametrize("batch,soft_v,res,overlap", [(4, 256, 8, False), (2, 1024, 20, True), (32, 1024, 20, True)])
def test_encoder_merged_path_equals_per_branch_path(batch, soft_v, res, overlap):
    """`ContactEncoder` with the branches merged (default) vs `merge_branches = False`.  Whole-stage node
    counts (1024-vertex meshes, 762-vertex spheres in even numbers): every output and gradient is the same
    bits; otherwise (the small case) the dW sums differ in their summation order only."""
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(batch, soft_vertices=soft_v, sphere_resolution=res))
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    enc.overlap_branches = overlap
    enc.merge_branches = True
    assert enc._mergeable(rest.x, rig.x)
    g_rest = torch.randn(rest.x.shape[0], 256, device=DEV)
    g_rig = torch.randn(rig.x.shape[0], 256, device=DEV)
    a0, b0, gr0 = _encoder_run(enc, rest, rig, g_rest, g_rig, merged=False)
    a1, b1, gr1 = _encoder_run(enc, rest, rig, g_rest, g_rig, merged=True)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    exact = rest.x.shape[0] % 32 == 0 and rig.x.shape[0] % 32 == 0
    for name in gr0:
        if exact:
            assert torch.equal(gr0[name], gr1[name]), name
        else:
            assert rel_err(_np(gr1[name]), _np(gr0[name])) < 2e-6, name


def test_encoder_merged_path_with_direct_gradient_bucket_and_adam():
    """The product's training configuration: gradients accumulated straight into the flat bucket
    (`GradBucket(direct=True)`), FlatAdam - three steps merged vs per-branch end with the same parameters."""
    from deformcontact_amd import dp
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(16))     # 16 spheres = 12,192 rows: whole 32-row stages
    g_rest = torch.randn(rest.x.shape[0], 256, device=DEV)
    g_rig = torch.randn(rig.x.shape[0], 256, device=DEV)
    finals = []
    for merged in (False, True):
        torch.manual_seed(0)
        enc = ContactEncoder([21, 25], 256).to(DEV)
        enc.merge_branches = merged
        bucket = dp.GradBucket(enc.parameters(), direct=True)
        opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
        bucket.zero()
        clear_cache()
        for _ in range(3):
            a, b = enc(rest, rig)
            torch.autograd.backward([a, b], [g_rest, g_rig])
            opt.step()
        torch.cuda.synchronize()
        finals.append({n: p.detach().clone() for n, p in enc.named_parameters()})
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n


def test_encoder_merged_path_under_hipgraph_with_changing_batches():
    """One captured step (merged adjacency build inside) replayed on new batches vs eager per-branch steps."""
    batches = [tuple(b.to(DEV) for b in synth.make_batch(16, first_idx=16 * i)) for i in range(3)]
    rest, _, rig = (b.clone() for b in batches[0])
    g_rest = torch.randn(rest.x.shape[0], 256, device=DEV)
    g_rig = torch.randn(rig.x.shape[0], 256, device=DEV)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    outs = {}

    def step():
        enc.zero_grad(set_to_none=False)
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [g_rest, g_rig])
        return a, b
    enc.merge_branches = True
    clear_cache()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for p in enc.parameters():
            p.grad = torch.zeros_like(p)
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    clear_cache()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for p in enc.parameters():
            p.grad.zero_()
        sa, sb = step()
    for i, (r, _, q) in enumerate(batches):
        rest.x.copy_(r.x), rest.edge_index.copy_(r.edge_index), rig.x.copy_(q.x), rig.edge_index.copy_(q.edge_index)
        graph.replay()
        torch.cuda.synchronize()
        outs[i] = (sa.clone(), sb.clone(), {n: p.grad.clone() for n, p in enc.named_parameters()})
    enc.merge_branches = False
    for i, (r, _, q) in enumerate(batches):
        a, b, gr = _encoder_run(enc, r, q, g_rest, g_rig, merged=False)
        assert torch.equal(a, outs[i][0]) and torch.equal(b, outs[i][1])
        for n in gr:
            assert torch.equal(gr[n], outs[i][2][n]), n


def test_three_layer_encoder_merged_path_equals_per_branch_path():
    """encoder_layers = 3: TWO grouped layers in a row - the first writes its outputs as the part views of the second
    one's merged slab (padding rows stay zero rows) - against the per-branch path, bit for bit (16 samples: whole stages)."""
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(16, soft_vertices=256, sphere_resolution=8))
    assert rest.x.shape[0] % 32 == 0 and rig.x.shape[0] % 32 == 0
    torch.manual_seed(3)
    enc = ContactEncoder([21, 25], 256, encoder_layers=3).to(DEV)
    with torch.no_grad():
        for n_, p_ in enc.named_parameters():
            if n_.endswith(".bias"):
                p_.uniform_(0.5, 1.0)            # relu(b) of a padding row would be far from zero
    enc.merge_branches = True
    assert enc._mergeable(rest.x, rig.x)
    g_rest = torch.randn(rest.x.shape[0], 256, device=DEV)
    g_rig = torch.randn(rig.x.shape[0], 256, device=DEV)
    a0, b0, gr0 = _encoder_run(enc, rest, rig, g_rest, g_rig, merged=False)
    a1, b1, gr1 = _encoder_run(enc, rest, rig, g_rest, g_rig, merged=True)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    for name in gr0:
        assert torch.equal(gr0[name], gr1[name]), name


def test_loader_prepare_builds_the_merged_adjacency_and_first_layer_slabs():
    """`loaders.prepare_for(model)` on the merged path: one merged adjacency (its windows registered for the parts'
    edge_index) + the first-layer hop slabs, ahead of the step; the step then builds nothing and equals an unprepared one."""
    from deformcontact_amd import loaders
    from deformcontact_amd.graph import graph_index
    rest, deff, rig = (b.to(DEV) for b in synth.make_batch(2, soft_vertices=256, sphere_resolution=8))
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    enc.merge_branches = True
    clear_cache()
    with torch.no_grad():
        a0, b0 = enc(rest, rig)
    clear_cache()
    built = loaders.prepare_for(enc)((rest, deff, rig))
    assert len(built) == 1 and built[0].parts is not None
    w = graph_index(rest.edge_index, rest.x.shape[0])
    assert w.merged is built[0] and len(getattr(w, "_hop_cache", {})) == 1
    calls = []
    real = ops._build_input_slab
    ops._build_input_slab = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            a1, b1 = enc(rest, rig)
    finally:
        ops._build_input_slab = real
    assert not calls and torch.equal(a0, a1) and torch.equal(b0, b1)
