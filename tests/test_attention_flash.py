"""Flash-style attention forward (``dc_attn_flash_fwd``) against the reference's formula in float64
(/root/reference/models/model.py:13-21: softmax(head(x_soft) head(x_rigid)^T, dim=-1) x_rigid, unmasked, no
1/sqrt(d)) and against the blocked three-launch form it replaces (scores are bit-identical between the two, so
outputs and the row log-sum-exp agree to rounding and the blocked backward runs unchanged on the flash forward's lse)."""
import numpy as np
import pytest
import torch

from deformcontact_amd import attention
from deformcontact_amd.attention import attention_core
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _run(q, k, v, go, flash, monkeypatch, block=2048):
    monkeypatch.setattr(attention, "FLASH", flash)
    q, k, v = (t.detach().clone().requires_grad_() for t in (q, k, v))
    out = attention_core(q, k, v, block_rows=block)
    out.backward(go)
    return _np(out), _np(q.grad), _np(k.grad), _np(v.grad)


@pytest.mark.parametrize("ns,nr,scale", [(300, 130, 0.5), (128, 32, 0.5), (1, 1, 1.0), (1000, 777, 0.5),
                                         (4096, 3048, 0.5),       # keys padded to a multiple of 128
                                         (257, 33, 2.0),          # large un-normalised scores: near one-hot weights
                                         (200, 95, 0.05)])        # nearly uniform weights
def test_flash_forward_vs_float64_and_blocked_form(ns, nr, scale, monkeypatch):
    torch.manual_seed(ns + nr)
    q = torch.randn(ns, 256, device=DEV) * scale
    k = torch.randn(nr, 256, device=DEV) * scale
    v = torch.randn(nr, 256, device=DEV)
    go = torch.randn(ns, 256, device=DEV)
    got = _run(q, k, v, go, True, monkeypatch)
    blk = _run(q, k, v, go, False, monkeypatch)
    qd, kd, vd = (t.double().cpu().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(qd @ kd.t(), dim=-1) @ vd
    ref.backward(go.double().cpu())
    want = [ref.detach().numpy(), qd.grad.numpy(), kd.grad.numpy(), vd.grad.numpy()]
    # fp32 scores of magnitude |s| carry ulp(|s|) of absolute error = that much RELATIVE error in a weight: 5e-6 for
    # the usual ranges, and never further from float64 than twice the blocked form (which has the same scores)
    e_flash, e_blk = rel_err(got[0], want[0]), rel_err(blk[0], want[0])
    assert e_flash <= max(5e-6, 2 * e_blk), (e_flash, e_blk)
    for name, g, b, w in zip(("dq", "dk", "dv"), got[1:], blk[1:], want[1:]):
        assert rel_err(g, w) <= max(1e-5, 2 * rel_err(b, w)), name


def test_flash_forward_row_statistics_and_heavy_tailed_rows(monkeypatch):
    """lse against float64, with rows whose scores span hundreds of units (the reference has no 1/sqrt(d)) and keys
    that differ by orders of magnitude in norm (per-row power-of-two scaling of the key image)."""
    torch.manual_seed(5)
    ns, nr = 384, 200
    q = torch.randn(ns, 256, device=DEV)
    q[::7] *= 6.0
    k = torch.randn(nr, 256, device=DEV)
    k[::5] *= 1e-3
    k[3::11] *= 4.0
    v = torch.randn(nr, 256, device=DEV)
    v[:, ::9] *= 100.0
    monkeypatch.setattr(attention, "FLASH", True)
    out = _np(attention_core(q, k, v))
    monkeypatch.setattr(attention, "FLASH", False)
    blk = _np(attention_core(q, k, v))
    qd, kd, vd = (t.double().cpu() for t in (q, k, v))
    kc = kd - kd.mean(dim=0, keepdim=True)                 # attention_core centres the keys (same softmax)
    s = qd @ kc.t()
    ref = (torch.softmax(s, dim=-1) @ vd).numpy()
    # scores of a few hundred carry ulp(|s|) ~ 3e-5 of absolute error in fp32 whoever computes them: the yardstick is
    # the blocked form, which has bit-identical scores
    e_flash, e_blk = rel_err(out, ref), rel_err(blk, ref)
    assert e_flash <= max(5e-6, 2 * e_blk), (e_flash, e_blk)
    # every output column separately (the V^T image is scaled per column)
    col = lambda a: (np.abs(a - ref).max(axis=0) / np.abs(ref).max(axis=0)).max()
    assert col(out) <= max(2e-5, 2 * col(blk)), (col(out), col(blk))


def test_flash_forward_is_deterministic_and_handles_ragged_query_tiles(monkeypatch):
    monkeypatch.setattr(attention, "FLASH", True)
    torch.manual_seed(1)
    q = torch.randn(515, 256, device=DEV) * 0.3            # 4 full tiles of 128 + 3 rows
    k = torch.randn(70, 256, device=DEV) * 0.3
    v = torch.randn(70, 256, device=DEV)
    a = _np(attention_core(q, k, v))
    b = _np(attention_core(q, k, v))
    assert np.array_equal(a, b)
    # a query's output does not depend on the other queries of its tile
    c = _np(attention_core(q[100:231], k, v))
    assert np.array_equal(a[100:231], c)
