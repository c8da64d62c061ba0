"""The 128 x 256 forms of the forward-shaped fp16x2 dense block - `k_fwd_h2d` (both operands by LDS-DMA, waves split by
role: the default for the wide layers) and `k_fwd_h2w` (every wave stages through registers: split reductions, the exp
epilogue, the correction operand, outputs without 16-byte rows; DC_H2_DMA=0 sends everything there) - here forced onto
small shapes (DC_H2_WIDE_MIN_TILES=1): bit-identical to the 64/128 x 128 kernel (`k_fwd_h2`) and within 2e-6 per row of
float64, on ragged shapes (rows / columns that do not fill a tile, odd stage counts, a single stage).

The tile shape is chosen from environment variables read once per process, so every variant runs in a child
process (which also keeps the selection of THIS process - the default - untouched)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array

import os
dev = torch.device("cuda:0")
L = _lib.lib()
gen = torch.Generator().manual_seed(1234)
worst = 0.0
_lib.kernel_trace(True)
for n, k, fo, relu, bias_on, ldpad in ((1000, 96, 256, True, True, 0), (4112, 1024, 256, True, True, 64),
                                       (515, 64, 320, False, True, 4), (256, 32, 256, False, False, 0),
                                       (2048, 160, 200, True, False, 8), (33, 1024, 17, True, True, 0),
                                       (128, 2048, 512, True, True, 0), (1, 32, 4, False, True, 0)):
    st = current_stream_ptr(dev)
    slab_base = (torch.rand(n, k + ldpad, generator=gen) * 4 - 2).to(dev)
    slab_base *= torch.logspace(-4, 0, n).unsqueeze(1).to(dev)            # rows on very different scales
    slab = slab_base[:, :k]
    w = ((torch.rand(fo, k, generator=gen) * 2 - 1) / k ** 0.5).to(dev)
    bias = (torch.rand(fo, generator=gen) - 0.5).to(dev) if bias_on else None
    rowmax = slab.abs().amax(1).contiguous()
    wmax = ops.weight_rowmax([w])
    wimg = torch.empty(fo, k, device=dev)
    _lib.check(L.dc_tag_weight_prep(_ptr_array([w]), 1, fo, k, wmax.data_ptr(), wimg.data_ptr(), None, None, st), "prep")
    out = torch.full((n, fo), float("nan"), device=dev)
    _lib.check(L.dc_tag_linear_fwd_h2p(slab.data_ptr(), slab.stride(0), wimg.data_ptr(),
                                       bias.data_ptr() if bias_on else None, int(relu), out.data_ptr(), fo, n, k, fo,
                                       rowmax.data_ptr(), wmax.data_ptr(), None, 0, st), "fwd_h2p")
    # the 64/128 x 128 kernel through the generic entry (fp32 weights, split in the kernel: never the wide shapes)
    ref32 = torch.empty(n, fo, device=dev)
    _lib.check(L.dc_tag_linear_fwd_h2(_ptr_array([slab]), _i64_array([slab.stride(0)]), _ptr_array([w]), 1,
                                      bias.data_ptr() if bias_on else None, int(relu), ref32.data_ptr(), fo, n, k, fo,
                                      rowmax.data_ptr(), wmax.data_ptr(), st), "fwd_h2")
    torch.cuda.synchronize()
    assert torch.equal(out, ref32), ("not bit-identical to k_fwd_h2", n, k, fo)
    ref = slab.double().cpu() @ w.double().cpu().t()
    if bias_on:
        ref = ref + bias.double().cpu()
    if relu:
        ref = ref.clamp_min(0)
    den = ref.abs().amax(1, keepdim=True).clamp_min(1e-300)
    err = float(((out.double().cpu() - ref).abs() / den).max())
    assert err < 2e-6, (err, n, k, fo)
    worst = max(worst, err)
counts = _lib.kernel_trace_counts()
_lib.kernel_trace(False)
want = os.environ.get("DC_EXPECT_KERNEL")
assert any(want in name for name in counts), (want, counts)
if want == "k_fwd_h2w":
    assert not any("k_fwd_h2d" in name for name in counts), counts
# split reduction on the wide tiles: a long K with few output tiles (attention weights x values)
for n, k, fo in ((256, 4096, 256), (200, 2080, 192)):
    st = current_stream_ptr(dev)
    x = (torch.rand(n, k, generator=gen) * 2 - 1).to(dev)
    w = ((torch.rand(fo, k, generator=gen) * 2 - 1) / k ** 0.5).to(dev)
    rowmax, wmax = x.abs().amax(1).contiguous(), ops.weight_rowmax([w])
    wimg = torch.empty(fo, k, device=dev)
    _lib.check(L.dc_tag_weight_prep(_ptr_array([w]), 1, fo, k, wmax.data_ptr(), wimg.data_ptr(), None, None, st), "prep")
    nb = L.dc_tag_linear_fwd_h2p_workspace_bytes(n, k, fo)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = torch.full((n, fo), float("nan"), device=dev)
    _lib.check(L.dc_tag_linear_fwd_h2p(x.data_ptr(), k, wimg.data_ptr(), None, 0, out.data_ptr(), fo, n, k, fo,
                                       rowmax.data_ptr(), wmax.data_ptr(), ws.data_ptr(), nb, st), "fwd_h2p split-K")
    torch.cuda.synchronize()
    ref = x.double().cpu() @ w.double().cpu().t()
    err = float(((out.double().cpu() - ref).abs() / ref.abs().amax(1, keepdim=True)).max())
    assert err < 2e-6, (err, n, k, fo)
    worst = max(worst, err)
print("OK", worst)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("dma", ["1", "0"])
def test_wide_dense_tiles_bit_identical_and_fp32_accurate(dma):
    env = dict(os.environ, DC_H2_WIDE="1", DC_H2_WIDE_MIN_TILES="1", DC_H2_DMA=dma, DC_H2_DMA_MIN_K="32", DC_EXPECT_KERNEL="k_fwd_h2d" if dma == "1" else "k_fwd_h2w")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK")


CHILD_DW = r"""
import hashlib
import sys
import torch
sys.path.insert(0, %(root)r)
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array

dev = torch.device("cuda:0")
L = _lib.lib()
gen = torch.Generator().manual_seed(99)
digest = hashlib.sha1()
worst = 0.0
for n, nseg in ((4096, 4), (2080, 1), (8192 + 32, 3)):
    fi = fo = 256
    st = current_stream_ptr(dev)
    slab = ops._alloc_slab(n, nseg * fi, dev)
    slab.copy_((torch.rand(n, nseg * fi, generator=gen) * 2 - 1).to(dev) * torch.logspace(-3, 0, n).unsqueeze(1).to(dev))
    g = ((torch.rand(n, fo, generator=gen) * 2 - 1) * torch.logspace(0, -3, n).unsqueeze(1)).to(dev)
    xs = [slab[:, s * fi:(s + 1) * fi] for s in range(nseg)]
    gws = [torch.empty(fo, fi, device=dev) for _ in range(nseg)]
    gb = torch.empty(fo, device=dev)
    nb = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi, fo, nseg)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    growmax, xrowmax = g.abs().amax(1).contiguous(), slab.abs().amax(1).contiguous()
    _lib.check(L.dc_tag_linear_bwd_dw_h2(g.data_ptr(), fo, None, fo, _ptr_array(xs), _i64_array([slab.stride(0)] * nseg),
                                         nseg, _ptr_array(gws), nseg, fi, gb.data_ptr(), 0, scratch.data_ptr(), nb,
                                         n, fi, fo, growmax.data_ptr(), xrowmax.data_ptr(), st), "dw_h2")
    torch.cuda.synchronize()
    for s in range(nseg):
        ref = g.double().cpu().t() @ xs[s].double().cpu()
        err = float((gws[s].double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (err, n, s)
        worst = max(worst, err)
        digest.update(gws[s].cpu().numpy().tobytes())
    refb = g.double().cpu().sum(0)
    assert float((gb.double().cpu() - refb).abs().max() / refb.abs().max()) < 2e-6
print("OK", worst, digest.hexdigest())
"""


@pytest.mark.gpu
def test_wide_dw_tiles_bit_identical_to_128x128_and_fp32_accurate():
    """`k_dw_h2w` (128 x 256 tiles, 32 nodes per stage; default for the wide layers) against float64, and
    bit for bit against `k_dw_split<2, false, 2>` (DC_DW_WIDE=0) - same chunks, same k order."""
    digests = []
    for wide in ("1", "0"):
        env = dict(os.environ, DC_DW_WIDE=wide)
        r = subprocess.run([sys.executable, "-c", CHILD_DW % {"root": ROOT}], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        last = r.stdout.strip().splitlines()[-1].split()
        assert last[0] == "OK"
        digests.append(last[2])
    assert digests[0] == digests[1]
