"""``dc_tag_weight_prep``: the transposed image of TALL matrices (the attention's keys / values handed over as "weights":
24,384 rows) is produced by coalesced column maxima + 64 x 64 tiles transposed through LDS (``k_wt_colmax`` / ``k_wt_image``)
instead of a wave per column; same scale, same rounding - checked bit for bit against the wave-per-column path (which
matrices below 2,048 rows still take) on the rows the two calls share, and against the definition."""
import numpy as np
import pytest
import torch

from deformcontact_amd import _lib
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _ptr_array

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def prep(ws, fo, fi):
    L = _lib.lib()
    nseg = len(ws)
    wmax = torch.empty(fo, device=DEV)
    wimg = torch.empty(fo, nseg * fi, device=DEV)
    wt = torch.empty(fi, nseg * fo, device=DEV)
    wtmax = torch.empty(fi, device=DEV)
    _lib.check(L.dc_tag_weight_prep(_ptr_array(ws), nseg, fo, fi, wmax.data_ptr(), wimg.data_ptr(), wt.data_ptr(),
                                    wtmax.data_ptr(), current_stream_ptr(torch.device(DEV))), "prep")
    torch.cuda.synchronize()
    return wmax, wimg, wt, wtmax


@pytest.mark.parametrize("fo,fi,nseg", [(4096, 256, 1), (2048 + 16, 208, 1), (3008, 64, 2), (24384, 256, 1)])
def test_tall_transposed_image_equals_the_wave_per_column_path_and_the_definition(fo, fi, nseg):
    _lib.kernel_trace(True)
    gen = torch.Generator().manual_seed(fo + fi)
    ws = [(torch.rand(fo, fi, generator=gen) * 2 - 1).to(DEV) for _ in range(nseg)]
    small = 2032                                            # below the tall threshold: the wave-per-column path
    for w in ws:                                            # every column's largest magnitude inside the shared rows
        w[:small] *= 1.0
        w[small:] *= 0.5
        w[5] = torch.where(torch.arange(fi, device=DEV) % 2 == 0, 1.5, -1.25)
    wmax, wimg, wt, wtmax = prep(ws, fo, fi)
    counts = _lib.kernel_trace_counts()
    _lib.kernel_trace(False)
    assert any("k_wt_image" in k for k in counts) and any("k_wt_colmax" in k for k in counts), counts
    # definition: column maxima exact; (h1 + h2) / scale within the split's 2^-22
    colmax = torch.stack([w.abs().amax(0) for w in ws]).amax(0)
    assert torch.equal(wtmax, colmax)
    img = wt.view(torch.float16).view(fi, nseg * fo // 16, 2, 16).float()          # [f][record][plane][16]
    rec = (img[:, :, 0] + img[:, :, 1]).reshape(fi, nseg * fo)
    e = ((torch.frexp(colmax)[1] - 1 + 127).clamp(15, 254)).float()               # biased exponent of the column maximum
    scale = torch.exp2(141.0 - e)[:, None]
    want = torch.cat([w.t() for w in ws], 1)
    assert float(((rec / scale) - want).abs().max()) <= 2.0 ** -21 * float(colmax.max())
    # the wave-per-column path on the first 2,032 rows: same column maxima -> the shared records are the same bytes
    ws_s = [w[:small].contiguous() for w in ws]
    _, _, wt_s, wtmax_s = prep(ws_s, small, fi)
    assert torch.equal(wtmax_s, wtmax)
    a = wt.view(torch.int16).view(fi, nseg, fo * 2)[:, :, :small * 2]
    b = wt_s.view(torch.int16).view(fi, nseg, small * 2)
    assert torch.equal(a, b)
