"""The fused row-wise passes of the GCNConv / GATConv layers (dc_gnn_epi.hip): PyG gcn_conv.py /
gat_conv.py's ``out = propagate(...) + bias``, the encoder's ReLU (/root/reference/models/model.py:71,77), GAT's
``alpha = (h * att).sum(-1)`` - each C entry against the unfused formulas, and the layers with the fusion on / off."""
import numpy as np
import pytest
import torch

from deformcontact_amd import _lib, ops
from deformcontact_amd import nn as dc_nn
from deformcontact_amd.graph import GraphIndex, clear_cache, current_stream_ptr
from tests.helpers import random_multigraph, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,e,f", [(1000, 7000, 256), (77, 300, 32), (513, 0, 64), (4099, 30000, 128)])
def test_aggregation_with_bias_and_relu_equals_three_passes_bitwise(n, e, f):
    ei = torch.from_numpy(random_multigraph(n, e, n + f)).to(DEV) if e else torch.zeros((2, 0), dtype=torch.int64, device=DEV)
    g = GraphIndex(ei, n, self_loops=True)
    torch.manual_seed(n)
    h, bias = torch.randn(n, f, device=DEV), torch.randn(f, device=DEV)
    ref = ops.hop(g.fwd, h)
    for b, relu in ((bias, True), (bias, False), (None, True), (None, False)):
        y = ops._agg_bias_act(g.fwd, g.fwd.w, h, b, relu)
        want = ref if b is None else ref + b
        want = torch.relu(want) if relu else want
        assert torch.equal(y, want), (b is not None, relu)


@pytest.mark.parametrize("n,f", [(1000, 256), (129, 32), (1, 64), (4099, 128)])
def test_mask_and_column_sums_in_one_pass(n, f):
    L = _lib.lib()
    torch.manual_seed(f)
    base = torch.randn(n, f + 8, device=DEV)
    gy, y = base[:, :f], torch.relu(torch.randn(n, f + 4, device=DEV))[:, :f]      # row-padded views
    outs = []
    for _ in range(2):
        gm = torch.empty(n, f, device=DEV)
        cs = torch.full((f,), 3.0, device=DEV)
        ws = torch.empty(L.dc_colsum_workspace_bytes(n, f, 1), dtype=torch.uint8, device=DEV)
        rc = L.dc_mask_colsum_f32(gy.data_ptr(), gy.stride(0), y.data_ptr(), y.stride(0), gm.data_ptr(), f, n, f,
                                  ws.data_ptr(), ws.numel(), cs.data_ptr(), 1, current_stream_ptr(torch.device(DEV)))
        assert rc == 0
        outs.append((gm.clone(), cs.clone()))
    want = gy * (y > 0)
    assert torch.equal(outs[0][0], want)
    assert torch.equal(outs[0][1], outs[1][1])                                         # deterministic
    assert rel_err(_np(outs[0][1]) - 3.0, want.double().sum(0).cpu().numpy()) < 2e-6  # accumulated onto the 3.0
    # no mask, no gm: plain column sums
    cs = torch.empty(f, device=DEV)
    ws = torch.empty(L.dc_colsum_workspace_bytes(n, f, 1), dtype=torch.uint8, device=DEV)
    assert L.dc_mask_colsum_f32(gy.data_ptr(), gy.stride(0), None, 0, None, 0, n, f, ws.data_ptr(), ws.numel(),
                                cs.data_ptr(), 0, current_stream_ptr(torch.device(DEV))) == 0
    assert rel_err(_np(cs), gy.double().sum(0).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("n,f", [(1000, 256), (130, 32), (4099, 64)])
def test_gat_attention_dot_products_forward_and_backward_vs_float64(n, f):
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    torch.manual_seed(n + f)
    h, a_s, a_d = torch.randn(n, f, device=DEV), torch.randn(f, device=DEV), torch.randn(f, device=DEV)
    a_src, a_dst = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    assert L.dc_gat_alpha_fwd(h.data_ptr(), f, a_s.data_ptr(), a_d.data_ptr(), a_src.data_ptr(), a_dst.data_ptr(), n, f, st) == 0
    h64 = h.double()
    assert rel_err(_np(a_src), (h64 @ a_s.double()).cpu().numpy()) < 2e-6
    assert rel_err(_np(a_dst), (h64 @ a_d.double()).cpu().numpy()) < 2e-6
    ga_s, ga_d = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    gh0 = torch.randn(n, f, device=DEV)
    gh, gs, gd = gh0.clone(), torch.empty(f, device=DEV), torch.empty(f, device=DEV)
    ws = torch.empty(L.dc_colsum_workspace_bytes(n, f, 2), dtype=torch.uint8, device=DEV)
    assert L.dc_gat_alpha_bwd(h.data_ptr(), f, ga_s.data_ptr(), ga_d.data_ptr(), a_s.data_ptr(), a_d.data_ptr(),
                              gh.data_ptr(), f, n, f, ws.data_ptr(), ws.numel(), gs.data_ptr(), gd.data_ptr(), 0, st) == 0
    want_gh = gh0.double() + ga_s.double()[:, None] * a_s.double() + ga_d.double()[:, None] * a_d.double()
    assert rel_err(_np(gh), want_gh.cpu().numpy()) < 2e-6
    assert rel_err(_np(gs), (ga_s.double() @ h64).cpu().numpy()) < 2e-6
    assert rel_err(_np(gd), (ga_d.double() @ h64).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("kind", ["GCNConv", "GATConv"])
@pytest.mark.parametrize("relu", [False, True])
def test_layers_with_and_without_the_fused_passes_agree(kind, relu, monkeypatch):
    n, e, fi, fo = 1500, 9000, 40, 256
    ei = torch.from_numpy(random_multigraph(n, e, 11)).to(DEV)
    torch.manual_seed(3)
    conv = getattr(dc_nn, kind)(fi, fo).to(DEV)
    with torch.no_grad():
        conv.bias.uniform_(-0.3, 0.3)
    x = torch.randn(n, fi, device=DEV, requires_grad=True)
    gy = torch.randn(n, fo, device=DEV)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_GNN_EPILOGUE", fused)
        clear_cache()
        conv.zero_grad(set_to_none=True)
        x.grad = None
        y = conv(x, ei, relu=relu)
        y.backward(gy)
        res.append([_np(y), _np(x.grad)] + [_np(p.grad) for p in conv.parameters()])
    names = ["y", "dx"] + [k for k, _ in conv.named_parameters()]
    for name, a, b in zip(names, *res):
        assert rel_err(a, b) < 3e-6, name
    if kind == "GCNConv":                                # same values exactly (the sum is finished before + bias); GAT's
        assert np.array_equal(res[0][0], res[1][0])      # attention dot products are summed in another order
