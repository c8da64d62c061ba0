"""`dc_tag_linear_fwd_narrow` (dc_dense_narrow.hip): the short-reduction forward block of the first encoder layers
(/root/reference/models/model.py:44-50: TAGConv(21, 256) / TAGConv(25, 256) -> PyG tag_conv.py `sum_k lins[k](x_k) + bias`,
then the F.relu of models/model.py:71,77).  Bit-identical to the six-product split kernel on packed weights
(`dc_tag_pack_weights` + `dc_tag_linear_fwd_split`), within 2e-6 per row of float64; ragged row counts,
all three padded widths, strided outputs (a column block of the next layer's slab)."""
import pytest
import torch

from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,fi,nseg,wpad,relu,bias_on,ldo_pad", [
    (32768, 21, 4, 96, True, True, 0),          # soft first layer at B = 32: four 32-row tiles per persistent workgroup
    (24384, 25, 4, 112, True, True, 832),       # rigid: 762 tiles (three rounds), output inside a wider slab
    (1000, 21, 4, 96, False, True, 0), (1, 25, 4, 112, True, False, 0), (65, 30, 4, 128, True, True, 64),
    (20000, 24, 4, 96, False, False, 0), (33, 7, 3, 96, True, True, 0)])
def test_narrow_forward_block_bit_identical_to_split_kernel_and_fp32_accurate(n, fi, nseg, wpad, relu, bias_on, ldo_pad):
    L = _lib.lib()
    dev = torch.device(DEV)
    st = current_stream_ptr(dev)
    fo = 256
    assert L.dc_tag_linear_fwd_narrow_ok(fi, nseg, wpad, fo) == 1
    gen = torch.Generator().manual_seed(n + fi)
    slab = torch.zeros(n, wpad + 8, device=dev)[:, :wpad]                 # leading dimension != width
    slab[:, :nseg * fi] = ((torch.rand(n, nseg * fi, generator=gen) * 4 - 2) *
                           torch.logspace(-3, 0, n).unsqueeze(1)).to(dev)
    ws = [((torch.rand(fo, fi, generator=gen) * 2 - 1) / fi ** 0.5).to(dev) for _ in range(nseg)]
    bias = (torch.rand(fo, generator=gen) - 0.5).to(dev) if bias_on else None
    base = torch.full((n, fo + ldo_pad), float("nan"), device=dev)
    out = base[:, :fo]
    _lib.check(L.dc_tag_linear_fwd_narrow(slab.data_ptr(), slab.stride(0), _ptr_array(ws), nseg, fi,
                                          bias.data_ptr() if bias_on else None, int(relu), out.data_ptr(), out.stride(0), n,
                                          wpad, fo, st), "narrow")
    wcat = torch.empty(fo, wpad, device=dev)
    _lib.check(L.dc_tag_pack_weights(_ptr_array(ws), nseg, wcat.data_ptr(), fo, fi, wpad, st), "pack")
    ref = torch.empty(n, fo, device=dev)
    _lib.check(L.dc_tag_linear_fwd_split(_ptr_array([slab]), _i64_array([slab.stride(0)]), _ptr_array([wcat]), 1,
                                         bias.data_ptr() if bias_on else None, int(relu), ref.data_ptr(), fo, n, wpad, fo, 6,
                                         st), "split")
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "not bit-identical to k_fwd_split<., 6>"
    if ldo_pad:
        assert torch.isnan(base[:, fo:]).all()                              # nothing written beside the output columns
    t = slab[:, :nseg * fi].double().cpu() @ torch.cat(ws, 1).double().cpu().t()
    if bias_on:
        t = t + bias.double().cpu()
    if relu:
        t = t.clamp_min(0)
    den = t.abs().amax(1, keepdim=True).clamp_min(1e-300)
    assert float(((out.double().cpu() - t).abs() / den).max()) < 2e-6


def test_first_layer_of_a_tagconv_uses_the_narrow_kernel_and_no_packing_launch():
    from deformcontact_amd import nn as dc_nn, synth
    rest, _, _ = synth.make_batch(2, soft_vertices=128, sphere_resolution=5)
    rest = rest.to(DEV)
    torch.manual_seed(0)
    conv = dc_nn.TAGConv(21, 256).to(DEV)
    _lib.kernel_trace(True)
    y = conv(rest.x, rest.edge_index, relu=True)
    torch.cuda.synchronize()
    _lib.kernel_trace(False)
    tr = _lib.kernel_trace_counts()
    assert any(k.startswith("k_fwd_narrow") for k in tr) and not any("k_pack_weights" in k or "k_fwd_split" in k for k in tr), tr
    keep, ops.NARROW_FWD = ops.NARROW_FWD, False
    try:
        y0 = conv(rest.x, rest.edge_index, relu=True)
    finally:
        ops.NARROW_FWD = keep
    assert torch.equal(y, y0)
    # input gradient of a narrow layer (not needed by the encoder; the one-segment dX block packs the weights itself)
    x = rest.x.clone().requires_grad_(True)
    conv(x, rest.edge_index, relu=True).sum().backward()
    gx = x.grad.clone()
    ops.NARROW_FWD = False
    try:
        x2 = rest.x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        conv(x2, rest.edge_index, relu=True).sum().backward()
    finally:
        ops.NARROW_FWD = keep
    assert torch.equal(gx, x2.grad)
