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


def _run(q, k, v, go, flash, monkeypatch, block=2048, flash_bwd=None, single=True):
    monkeypatch.setattr(attention, "FLASH", flash)
    monkeypatch.setattr(attention, "FLASH_BWD", flash if flash_bwd is None else flash_bwd)
    monkeypatch.setattr(attention, "FLASH_BWD_SINGLE", single)
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


@pytest.mark.parametrize("ns,nr", [(300, 130), (1, 1), (1000, 777), (4096, 3048), (515, 70), (5000, 130), (33, 64)])
def test_flash_backward_equals_blocked_backward_to_rounding_and_float64(ns, nr, monkeypatch):
    """Same forward (flash), backward through ``dc_attn_flash_ds`` + three large GEMMs vs the blocked six-launch form."""
    torch.manual_seed(3 * ns + nr)
    q = torch.randn(ns, 256, device=DEV) * 0.5
    k = torch.randn(nr, 256, device=DEV) * 0.5
    v = torch.randn(nr, 256, device=DEV)
    go = torch.randn(ns, 256, device=DEV)
    got = _run(q, k, v, go, True, monkeypatch, flash_bwd=True)
    two = _run(q, k, v, go, True, monkeypatch, flash_bwd=True, single=False)
    blk = _run(q, k, v, go, True, monkeypatch, flash_bwd=False)
    assert np.array_equal(got[0], blk[0]) and np.array_equal(two[0], blk[0])     # same forward
    qd, kd, vd = (t.double().cpu().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(qd @ kd.t(), dim=-1) @ vd
    ref.backward(go.double().cpu())
    for name, g, t, b, w in zip(("dq", "dk", "dv"), got[1:], two[1:], blk[1:], (qd.grad, kd.grad, vd.grad)):
        assert rel_err(g, w.numpy()) <= max(1e-5, 2 * rel_err(b, w.numpy())), ("single sweep", name)
        assert rel_err(t, w.numpy()) <= max(1e-5, 2 * rel_err(b, w.numpy())), ("two sweeps", name)
    # the sum over the keys of dK is mathematically zero (rows of dS sum to zero): the single-sweep form must keep it
    # as small as the two-sweep form does (this is what the eps correction of dK is for)
    kscale = np.abs(got[2]).sum(axis=0).max() + 1e-30
    assert np.abs(got[2].sum(axis=0)).max() <= 4 * np.abs(two[2].sum(axis=0)).max() + 2e-6 * kscale


def test_flash_ds_kernel_outputs_vs_float64():
    """``dc_attn_flash_ds`` itself: P, dS and the row maxima against float64; padded key columns exactly zero; the rows
    of dS sum to zero to rounding (delta is formed from the same P and dP)."""
    from deformcontact_amd import _lib
    from deformcontact_amd.graph import current_stream_ptr
    torch.manual_seed(11)
    ns, nr = 333, 150
    dev = torch.device(DEV)
    q = torch.randn(ns, 256, device=dev) * 0.4
    k = torch.randn(nr, 256, device=dev) * 0.4
    v = torch.randn(nr, 256, device=dev)
    go = torch.randn(ns, 256, device=dev)
    L, st = _lib.lib(), current_stream_ptr(dev)
    nrp = attention._ceil_keys(nr)
    kp, vp = attention._pad_rows(k, nrp), attention._pad_rows(v, nrp)
    kmax, kimg, _, _ = attention._prep(L, kp, False, st)
    vmax, vimg, _, _ = attention._prep(L, vp, False, st)
    kuns, vuns = torch.empty_like(kmax), torch.empty_like(vmax)
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, kmax.data_ptr(), kuns.data_ptr(), st), "prep")
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, vmax.data_ptr(), vuns.data_ptr(), st), "prep")
    qmax, gomax = attention._rowabsmax(L, q, st), attention._rowabsmax(L, go, st)
    qd, kd, vd, god = (t.double().cpu() for t in (q, k, v, go))
    s64 = qd @ kd.t()
    lse64 = torch.logsumexp(s64, dim=1)
    lse = lse64.float().to(dev)
    p = torch.full((ns, nrp), 7.0, device=dev)
    ds = torch.full((ns, nrp), 7.0, device=dev)
    dsmax = torch.empty(ns, device=dev)
    _lib.check(L.dc_attn_flash_ds(q.data_ptr(), 256, qmax.data_ptr(), go.data_ptr(), 256, gomax.data_ptr(),
                                  kimg.data_ptr(), kuns.data_ptr(), vimg.data_ptr(), vuns.data_ptr(), lse.data_ptr(),
                                  ns, nr, nrp, 256, p.data_ptr(), ds.data_ptr(), nrp, dsmax.data_ptr(), None, None, st), "ds")
    p64 = torch.exp(s64 - lse64[:, None])
    dp64 = god @ vd.t()
    ds64 = p64 * (dp64 - (p64 * dp64).sum(dim=1, keepdim=True))
    assert np.all(_np(p)[:, nr:] == 0) and np.all(_np(ds)[:, nr:] == 0)
    assert rel_err(_np(p)[:, :nr], p64.numpy()) < 5e-6
    assert rel_err(_np(ds)[:, :nr], ds64.numpy()) < 1e-5
    assert np.allclose(_np(dsmax), np.abs(_np(ds)).max(axis=1), rtol=0, atol=0)
    rows = np.abs(_np(ds).astype(np.float64).sum(axis=1))
    assert rows.max() <= 2e-6 * np.abs(_np(ds)).sum(axis=1).max()


def test_one_sweep_backward_when_the_values_share_a_large_common_component(monkeypatch):
    """Post-ReLU features: every value row = a large common vector + a small individual part, so dP = dO V^T is nearly
    constant over the keys and |delta_i| is ~100x the spread of dP_i. - the regime in which an inconsistent delta shows
    (round 2's precision fix).  The one-sweep backward (delta = rowsum(dO * O), eps correction in dK AND dQ) must stay
    as close to float64 as the two-sweep form."""
    torch.manual_seed(21)
    ns, nr = 700, 260
    q = torch.randn(ns, 256, device=DEV) * 0.4
    k = torch.randn(nr, 256, device=DEV) * 0.4 + 1.5                 # keys with a common component as well
    v = 10.0 + 0.1 * torch.randn(nr, 256, device=DEV)
    go = torch.randn(ns, 256, device=DEV)
    one = _run(q, k, v, go, True, monkeypatch, flash_bwd=True, single=True)
    two = _run(q, k, v, go, True, monkeypatch, flash_bwd=True, single=False)
    qd, kd, vd = (t.double().cpu().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(qd @ kd.t(), dim=-1) @ vd
    ref.backward(go.double().cpu())
    for name, a, b, w in zip(("dq", "dk", "dv"), one[1:], two[1:], (qd.grad, kd.grad, vd.grad)):
        ea, eb = rel_err(a, w.numpy()), rel_err(b, w.numpy())
        assert ea <= max(1e-5, 2 * eb), (name, ea, eb)
    # and the head-weight-like reduction that carried the bias: sum over the keys of dK, sum over the queries of dQ
    for a, b, w in ((one[2], two[2], kd.grad), (one[1], two[1], qd.grad)):
        sa, sb, sw = a.sum(axis=0), b.sum(axis=0), w.numpy().sum(axis=0)
        scale = np.abs(w.numpy()).sum(axis=0).max()
        assert np.abs(sa - sw).max() <= max(2 * np.abs(sb - sw).max(), 2e-6 * scale)


@pytest.mark.parametrize("ns,nr", [(4 * 256, 4 * 114), (3 * 200 + 7, 5 * 61)])
def test_cross_attention_host_paths_agree_with_float64(ns, nr):
    """`graphnet.CrossAttention` on the stock-attention path (the shipped batch 4): per head and per graph / the heads' shared
    Linear once over the rows of both graphs (`joint_max_rows`) / all heads through one dense block with batched score and
    pooling products (`batched_heads`) - three launch structures of one function (`models/model.py:7-21`): outputs and every
    gradient within 1e-5 of a float64 evaluation of the reference formula, scaled by the tensor's largest element."""
    from deformcontact_amd.graphnet import CrossAttention
    torch.manual_seed(3)
    d, heads = 256, 2
    att = CrossAttention(d, heads).to(DEV)
    xs = (torch.rand(ns, d, device=DEV) * (torch.rand(ns, d, device=DEV) < 0.5)).requires_grad_(True)   # post-ReLU-like rows
    xr = (torch.rand(nr, d, device=DEV) * (torch.rand(nr, d, device=DEV) < 0.5)).requires_grad_(True)
    with torch.no_grad():
        for h in att.attention_heads:
            h.weight.mul_(0.35)                                            # scores of a few units: softmax neither flat nor one-hot
    go = torch.randn(ns, heads * d, device=DEV)

    def run(module, a, b, g):
        out = module(a, b)
        grads = torch.autograd.grad(out, [a, b] + list(module.parameters()), g)
        return [out] + list(grads)

    att64 = CrossAttention(d, heads).double()
    att64.load_state_dict({k: v.detach().double().cpu() for k, v in att.state_dict().items()})
    ref = run(att64, xs.detach().double().cpu().requires_grad_(True), xr.detach().double().cpu().requires_grad_(True),
              go.double().cpu())
    names = ["out", "d x_soft", "d x_rigid"] + [f"d {k}" for k, _ in att.named_parameters()]
    results = {}
    for label, joint, batched in (("per graph", 0, False), ("joint linear", 16384, False), ("batched heads", 16384, True)):
        att.fused, att.joint_max_rows, att.batched_heads = "0", joint, batched
        got = run(att, xs, xr, go)
        results[label] = got
        for name, r, gt in zip(names, ref, got):
            scale = float(r.detach().abs().max())
            err = float((gt.detach().double().cpu() - r.detach()).abs().max()) / max(scale, 1e-30)
            assert err <= 1e-5, f"{label}: {name} is {err:.2e} of its scale from float64"
    for name, a, b in zip(names, results["per graph"], results["batched heads"]):
        assert float((a - b).detach().abs().max()) <= 4e-6 * float(a.detach().abs().max()), name
