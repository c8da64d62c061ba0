"""BASELINE.json configs[4] is forward + BACKWARD: the bf16-storage TAGConv layer's gradients.
`dc_tag_linear_bwd_dw_bf16` alone against float64, a two-layer bf16 stack (slab hand-off, fp32 master weights) against
the float64 evaluation of the same layers, and one layer forward + backward on the 100k-point radius graph against the
float64 closed forms.  Stated bf16 tolerance (as for the forward, tests/test_gpu_parity.py): 2e-2 of max |ref|, 4e-3 rms
for quantities that pass through bf16 storage; weight gradients (fp32 sums over the nodes of exact bf16 x bf16
products) 4e-3.  Reference call site: /root/reference/utils/pointcloud_utils.py:7-13 (the radius graph), PyG TAGConv
under torch.autocast(bfloat16)."""
import numpy as np
import pytest
import torch

import deformcontact_amd as dc
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import GraphIndex, current_stream_ptr
from deformcontact_amd.ops import _ptr_array
from oracle import pyg_ref
from oracle.weights import hashed_uniform
from tests.helpers import random_multigraph, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().float().cpu().numpy() if t.dtype == torch.bfloat16 else t.detach().cpu().numpy()


@pytest.mark.parametrize("n,fi,fo,nseg,bias", [(4096, 256, 256, 4, True), (1003, 256, 128, 1, True), (31, 256, 128, 2, False),
                                               (20000, 512, 256, 2, True), (1, 256, 128, 1, True)])
def test_dw_bf16_kernel_vs_float64(n, fi, fo, nseg, bias):
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    gen = torch.Generator().manual_seed(n + fi)
    slab = torch.randn(n, nseg * fi + 64, generator=gen).to(DEV).bfloat16()[:, :nseg * fi]     # a column slice
    g = torch.randn(n, fo + 8, generator=gen).to(DEV).bfloat16()[:, :fo]
    gws = [torch.full((fo, fi), 7.0, device=DEV) for _ in range(nseg)]
    gb = torch.full((fo,), 7.0, device=DEV) if bias else None
    nb = L.dc_tag_linear_bwd_dw_bf16_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)

    def run(acc):
        _lib.check(L.dc_tag_linear_bwd_dw_bf16(g.data_ptr(), g.stride(0), slab.data_ptr(), slab.stride(0), nseg,
                                               _ptr_array(gws), gb.data_ptr() if bias else None, acc,
                                               scratch.data_ptr(), nb, n, fi, fo, st), "dw")
    run(0)
    g64, x64 = g.double().cpu(), slab.double().cpu()
    for s in range(nseg):
        ref = g64.t() @ x64[:, s * fi:(s + 1) * fi]
        assert rel_err(_np(gws[s]), ref.numpy()) < 2e-6, s
    if bias:
        assert rel_err(_np(gb), g64.sum(0).numpy()) < 2e-6
    first = [w.clone() for w in gws]
    run(1)                                                                    # accumulate into the outputs
    for s in range(nseg):
        assert torch.allclose(gws[s], 2 * first[s], rtol=1e-6, atol=0)


@pytest.mark.parametrize("n,fi,fo,nseg,bias", [(4096, 256, 256, 4, True), (1003, 256, 128, 1, True), (31, 256, 128, 2, False),
                                               (100000, 256, 256, 4, True), (1, 256, 128, 1, True)])
def test_dw_bf16_by_lds_dma_equals_the_register_staged_kernel_bitwise(n, fi, fo, nseg, bias, monkeypatch):
    """k_dw_bf16d (both operands by LDS-DMA, dense swizzled images, loading waves) against k_dw_bf16: the same
    products in the same order - identical weight and bias gradients, ragged last stages and chunks included."""
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    gen = torch.Generator().manual_seed(n + fo)
    slab = torch.randn(n, nseg * fi + 64, generator=gen).to(DEV).bfloat16()[:, :nseg * fi]
    g = torch.randn(n, fo + 8, generator=gen).to(DEV).bfloat16()[:, :fo]
    nb = L.dc_tag_linear_bwd_dw_bf16_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    got = {}
    for mode, kernel in (("0", "k_dw_bf16"), ("1", "k_dw_bf16d")):
        monkeypatch.setenv("DC_DW_BF16_DMA", mode)
        gws = [torch.full((fo, fi), 7.0, device=DEV) for _ in range(nseg)]
        gb = torch.full((fo,), 7.0, device=DEV) if bias else None
        scratch.fill_(0xFF)
        _lib.kernel_trace(True)
        _lib.check(L.dc_tag_linear_bwd_dw_bf16(g.data_ptr(), g.stride(0), slab.data_ptr(), slab.stride(0), nseg,
                                               _ptr_array(gws), gb.data_ptr() if bias else None, 0,
                                               scratch.data_ptr(), nb, n, fi, fo, st), "dw")
        _lib.kernel_trace(False)
        names = _lib.kernel_trace_counts()
        assert any(k.split("(")[0].endswith(kernel) for k in names), (kernel, sorted(names))
        got[mode] = (gws, gb)
    for a, b in zip(got["0"][0], got["1"][0]):
        assert torch.equal(a, b)
    if bias:
        assert torch.equal(got["0"][1], got["1"][1])


@pytest.mark.parametrize("n,f,off,g_bf16,m_bf16,kernel", [
    (1000, 256, 0, False, True, "k_mask_grad_bf16x8<false, true>"), (1000, 256, 8, True, True, "k_mask_grad_bf16x8<true, true>"),
    (333, 64, 0, False, False, "k_mask_grad_bf16x8<false, false>"), (333, 64, 8, True, False, "k_mask_grad_bf16x8<true, false>"),
    (1000, 256, 4, False, True, "k_mask_grad_bf16"),              # a column slice off the 16-byte grid: scalar kernel
    (77, 20, 0, False, False, "k_mask_grad_bf16"), (5, 8, 0, True, True, "k_mask_grad_bf16x8<true, true>")])
def test_mask_grad_bf16_is_the_rounded_masked_gradient(n, f, off, g_bf16, m_bf16, kernel):
    """dc_tag_mask_grad_bf16 = bf16(g * (out > 0)) element by element (RNE), through the 8-column kernel and the
    scalar one (model.py:69-78: the ReLU between the layers, backward)."""
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    gen = torch.Generator().manual_seed(n + f + off)
    gfull = torch.randn(n, f + 16, generator=gen).to(DEV)
    ofull = torch.randn(n, f + 16, generator=gen).to(DEV)
    ofull[ofull.abs() < 0.3] = 0.0                                          # exact zeros: masked
    if g_bf16:
        gfull = gfull.bfloat16()
    if m_bf16:
        ofull = ofull.bfloat16()
    g, o = gfull[:, off:off + f], ofull[:, off:off + f]
    gm = torch.full((n, f + 8), 3.0, device=DEV).bfloat16()
    _lib.kernel_trace(True)
    _lib.check(L.dc_tag_mask_grad_bf16(g.data_ptr(), g.stride(0), int(g_bf16), o.data_ptr(), o.stride(0), int(m_bf16),
                                       gm.data_ptr(), gm.stride(0), n, f, st), "mask")
    _lib.kernel_trace(False)
    names = _lib.kernel_trace_counts()
    norm = lambda k: k.replace(" ", "").replace("(", "").replace(")", "").replace("dc::", "")   # noqa: E731
    assert [norm(k) for k in names] == [norm(kernel)], sorted(names)
    want = torch.where(o.float() > 0, g.float(), torch.zeros((), device=DEV)).bfloat16()
    assert torch.equal(gm[:, :f], want)
    assert torch.all(gm[:, f:] == 3.0)                                      # nothing written past F


def _double_stack(convs, x, ei, masks=None):
    """float64 evaluation of a stack of TAGConv + ReLU layers on bf16-rounded weights (masks: the HIP path's)."""
    refs = []
    for c in convs:
        r = pyg_ref.TAGConv(c.in_channels, c.out_channels).double()
        r.load_state_dict({k: v.detach().cpu().bfloat16().double() if "lins" in k else v.detach().cpu().double()
                           for k, v in c.state_dict().items()})
        refs.append(r)
    h = x
    for i, r in enumerate(refs):
        pre = r(h, ei)
        h = pre * masks[i] if masks is not None else torch.relu(pre)
    return refs, h


def test_two_layer_bf16_stack_forward_backward_vs_float64():
    n, f = 1500, 256
    ei = torch.from_numpy(random_multigraph(n, 9000, 31)).to(DEV)
    x = torch.from_numpy(hashed_uniform((n, f), 3, 2.0)).to(DEV).bfloat16().requires_grad_(True)
    torch.manual_seed(1)
    c1, c2 = dc.nn.TAGConv(f, f).to(DEV), dc.nn.TAGConv(f, f).to(DEV)
    with torch.no_grad():
        for c in (c1, c2):
            c.bias.uniform_(-0.2, 0.2)
    h = c1(x, ei, relu=True, next_conv=c2)
    y = c2(h, ei, relu=True)
    assert y.dtype == torch.bfloat16 and h._base is not None                 # slab hand-off between the layers
    gy = torch.from_numpy(hashed_uniform((n, f), 9, 1.0)).to(DEV).bfloat16()
    y.backward(gy)
    assert x.grad is not None and x.grad.dtype == torch.bfloat16 and c1.lins[0].weight.grad.dtype == torch.float32
    x64 = x.detach().double().cpu().requires_grad_(True)
    masks = [(h.detach().cpu() > 0).double(), (y.detach().cpu() > 0).double()]
    refs, y64 = _double_stack((c1, c2), x64, ei.cpu(), masks)
    y64.backward(gy.double().cpu())
    for name, got, ref, tol in [("y", y, y64, 2e-2), ("gx", x.grad, x64.grad, 3e-2)]:
        assert rel_err(_np(got), ref.detach().numpy()) < tol, name
    for c, r in zip((c1, c2), refs):
        for (name, p), q in zip(c.named_parameters(), r.parameters()):
            assert rel_err(_np(p.grad), q.grad.numpy()) < 1.5e-2, name        # gm / activations went through bf16


def test_config4_radius100k_bf16_tagconv_forward_backward_vs_float64():
    """configs[4] at full size: TAGConv(256, 256) + ReLU on the 100k-point radius graph, bf16 features, forward AND
    backward, against the float64 closed forms on the same bf16-rounded inputs (mask = the HIP path's own)."""
    import scipy.sparse as sp
    from deformcontact_amd import synth
    pos, ei = synth.radius_graph_points(100_000, radius=0.02, max_num_neighbors=32)
    ei = ei.to(DEV)
    n, f = pos.shape[0], 256
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, f, generator=gen).to(DEV).bfloat16().requires_grad_(True)
    g = torch.randn(n, f, generator=gen).to(DEV).bfloat16()
    torch.manual_seed(5)
    conv = dc.nn.TAGConv(f, f).to(DEV)
    with torch.no_grad():
        conv.bias.copy_(torch.from_numpy(hashed_uniform((f,), 5, 0.3)))
    y = conv(x, ei, relu=True)
    y.backward(g)
    torch.cuda.synchronize()
    ei_c = ei.cpu().numpy()
    deg = np.bincount(ei_c[1], minlength=n).astype(np.float64)
    dis = np.zeros_like(deg)
    dis[deg > 0] = deg[deg > 0] ** -0.5
    a = sp.csr_matrix((dis[ei_c[0]] * dis[ei_c[1]], (ei_c[1], ei_c[0])), shape=(n, n))
    at = a.T.tocsr()
    ws = [lin.weight.detach().bfloat16().double().cpu().numpy() for lin in conv.lins]
    xs = [x.detach().double().cpu().numpy()]
    for _ in range(3):
        xs.append(a @ xs[-1])
    ref = sum(xk @ w.T for xk, w in zip(xs, ws)) + conv.bias.detach().double().cpu().numpy()
    mask = (y.detach().float().cpu().numpy() > 0)
    got = y.detach().double().cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref * mask).max() / scale < 2e-2
    gm = g.double().cpu().numpy() * mask
    gs = [gm]
    for _ in range(3):
        gs.append(at @ gs[-1])
    gx_ref = sum(gk @ w for gk, w in zip(gs, ws))
    gx = x.grad.double().cpu().numpy()
    sx = np.abs(gx_ref).max()
    assert np.abs(gx - gx_ref).max() / sx < 2e-2 and np.sqrt(np.mean((gx - gx_ref) ** 2)) / sx < 4e-3
    for k, lin in enumerate(conv.lins):
        dw_ref = gm.T @ xs[k]
        assert rel_err(_np(lin.weight.grad), dw_ref) < 4e-3, f"dW_{k}"         # x_k went through k bf16 roundings
    assert rel_err(_np(conv.bias.grad), gm.sum(0)) < 1e-5
