"""Blocked cross-attention on the fp16x2 dense kernels.

The reference's ``MultiHeadAttention`` (``/root/reference/models/model.py:7-21``) computes, per
head, ``softmax(head(x_soft) @ head(x_rigid).T, dim=-1) @ x_rigid`` - unmasked over the whole
batch, no ``1/sqrt(d)`` - by materialising the ``[N_s, N_r]`` score matrix and keeping the softmax
weights for autograd: 2 x 3.2 GB per head at batch 32.  ``attention_core`` computes the same
function with the score matrix alive only for a block of ``block_rows`` soft rows:

  forward   per block:  S = Q_b K^T  ->  P = softmax(S) (row log-sum-exp kept)  ->  O_b = P V
  backward  per block:  S, P recomputed;  dP = dO_b V^T;  dS = P * (dP - rowsum(dO_b * O_b));
                        dQ_b = dS K;  dK += dS^T Q_b;  dV += P^T dO_b

Every GEMM runs on the library's fp16x2 kernels (``dc_tag_linear_fwd_h2p`` with the right-hand
operand pre-split by ``dc_tag_weight_prep``; ``dc_tag_linear_bwd_dw_h2`` for the two products that
contract over the soft rows), the row-wise parts on ``dc_attn_*``; fp32 storage throughout, error
at the level of fp32 accumulation.  Activation memory is ``block_rows x N_r`` floats (x2 in the
backward) instead of ``N_s x N_r`` per head.
"""
from __future__ import annotations

import torch

from . import _lib
from .graph import _require_cuda, current_stream_ptr
from .ops import _i64_array, _ptr_array

import os

#: soft rows per block: the block's score matrix (x2 in the backward) should stay in the 256 MiB
#: Infinity Cache between the GEMM that writes it and the row kernel / GEMM that reads it
BLOCK_ROWS = 2048


def _ceil16(n: int) -> int:
    return (n + 15) // 16 * 16


def _ceil_keys(n: int) -> int:
    """Padded key count: a multiple of 16 (one MFMA k-step); of 128 once there are enough keys, so that the dW-shaped
    products of the backward (dK, dV: [keys, d]) tile into the 128 x 256 blocks of ``k_dw_h2w``."""
    return (n + 127) // 128 * 128 if n >= 1024 else (n + 31) // 32 * 32      # (32: one key tile of the flash forward)


def _pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    if t.size(0) == rows and t.is_contiguous():
        return t
    out = torch.zeros((rows, t.size(1)), dtype=t.dtype, device=t.device)
    out[:t.size(0)].copy_(t)
    return out


def _prep(L, mat: torch.Tensor, want_t: bool, st):
    """``dc_tag_weight_prep`` over one matrix [R, C]: (rowmax [R], image [R, C], timage [C, R] | None,
    trowmax [C] | None) - the scaled fp16x2 images of its rows and of its columns."""
    r, c = mat.shape
    dev = mat.device
    rowmax = torch.empty(r, dtype=torch.float32, device=dev)
    img = torch.empty((r, c), dtype=torch.float32, device=dev)
    timg = torch.empty((c, r), dtype=torch.float32, device=dev) if want_t else None
    tmax = torch.empty(c, dtype=torch.float32, device=dev) if want_t else None
    _lib.check(L.dc_tag_weight_prep(_ptr_array([mat]), 1, r, c, rowmax.data_ptr(), img.data_ptr(),
                                    timg.data_ptr() if want_t else None,
                                    tmax.data_ptr() if want_t else None, st), "dc_tag_weight_prep")
    return rowmax, img, timg, tmax


def _rowabsmax(L, t: torch.Tensor, st) -> torch.Tensor:
    out = torch.empty(t.size(0), dtype=torch.float32, device=t.device)
    _lib.check(L.dc_rowabsmax_f32(t.data_ptr(), t.stride(0), t.size(0), t.size(1), out.data_ptr(), st),
               "dc_rowabsmax_f32")
    return out


#: forward in ONE launch per head (``dc_attn_flash_fwd``: online softmax over 32-key tiles, the scores never leave the
#: compute unit) when d = dv = 256 - the shipped hidden width; ``DC_ATTN_FLASH=0``: the blocked three-launch form.
#: Scores are bit-identical between the two (same products, same order), outputs agree to rounding.
FLASH = os.environ.get("DC_ATTN_FLASH", "1") != "0"
FLASH_D = 256
#: backward: P and dS for ALL rows from one two-sweep launch (``dc_attn_flash_ds``) followed by three large GEMMs,
#: instead of six launches per 2,048-row block; needs 2 x Ns x Nr floats (6.4 GB per head at batch 32).
#: ``DC_ATTN_FLASH_BWD=0``: the blocked backward.
FLASH_BWD = os.environ.get("DC_ATTN_FLASH_BWD", "1") != "0"
#: above this many bytes of P + dS the backward falls back to the blocked form
FLASH_BWD_MAX_BYTES = int(float(os.environ.get("DC_ATTN_FLASH_BWD_MAX_GB", "16")) * (1 << 30))


def _flash_bwd_budget(dev) -> int:
    """Bytes the P + dS matrices of the one-launch backward may take: the configured cap, and never more than half
    of what the device (driver + this process' caching allocator) can still hand out - beyond it the blocked backward
    (1.4 GiB at batch 32) runs instead of an out-of-memory error.  Under stream capture the query is
    skipped (it would be baked in anyway): the cap alone decides."""
    if torch.cuda.is_current_stream_capturing():
        return FLASH_BWD_MAX_BYTES
    free, _ = torch.cuda.mem_get_info(dev)
    st = torch.cuda.memory_stats(dev)
    cached = st.get("reserved_bytes.all.current", 0) - st.get("allocated_bytes.all.current", 0)
    return min(FLASH_BWD_MAX_BYTES, (free + max(cached, 0)) // 2)
#: one sweep instead of two in ``dc_attn_flash_ds``: dS' is formed with delta = rowsum(dO * O), the kernel returns by how
#: much the consistent delta differs (eps), and dK - the product that is sensitive to rows of dS not summing to zero -
#: takes dS' - eps o P at load time (``dc_tag_linear_bwd_dw_h2_corr``).  ``DC_ATTN_FLASH_BWD_SINGLE=0``: two sweeps.
FLASH_BWD_SINGLE = os.environ.get("DC_ATTN_FLASH_BWD_SINGLE", "1") != "0"
#: (one-sweep form: dQ = (dS' - eps o P) K as well, ``dc_tag_linear_fwd_h2p_corr``: what eps adds to an uncorrected dQ_i is
#: eps_i times the attention-weighted mean of the centred keys - small next to dQ_i unless dP is nearly constant over the keys,
#: which is exactly what post-ReLU features look like.  The blocked backward recomputes a block's weights as exp(s - lse) in
#: the score GEMM's epilogue, ``dc_tag_linear_fwd_h2p_exp``.)


def _gemm(L, x, ldx, rows, k, img, fo, out, ldo, xmax, wmax, st, ws=None):
    """out[rows, fo] = x[rows, k] . W^T with W given as its pre-split image (fp16x2, LDS-DMA);
    ``ws``: workspace that lets a long reduction with a small output be cut into ranges."""
    _lib.check(L.dc_tag_linear_fwd_h2p(x, ldx, img.data_ptr(), None, 0, out, ldo, rows, k, fo, xmax,
                                       wmax.data_ptr(), ws.data_ptr() if ws is not None else None,
                                       ws.numel() if ws is not None else 0, st), "dc_tag_linear_fwd_h2p")


#: delta_i = sum_j P_ij dP_ij / sum_j P_ij is formed inside dc_attn_ds_rows / dc_attn_flash_ds, not as rowsum(dO * O): the
#: latter leaves sum_j dS_ij ~ 3e-7 |delta_i| (the recomputed weights exp(S - lse) do not sum to exactly 1), a bias that
#: survives into sum_j dK_j - mathematically zero - and cost 1e-5 .. 4e-5 in the shared head weights against float64
#: (r02 attn_diag: 1.9e-5 -> 2.2e-6).


def _splitk_ws(L, rows, k, fo, dev):
    nb = L.dc_tag_linear_fwd_h2p_workspace_bytes(rows, k, fo)
    return torch.empty(nb, dtype=torch.uint8, device=dev) if nb > 0 else None


class _AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, block_rows: int):
        for name, t in (("q", q), ("k", k), ("v", v)):
            _require_cuda(t, name)
            if t.dim() != 2 or t.dtype != torch.float32:
                raise ValueError(f"attention_core: {name} must be a 2-D float32 tensor")
        ns, d = q.shape
        nr = k.size(0)
        if k.size(1) != d or v.size(0) != nr or d % 16 != 0:
            raise ValueError("attention_core: q [Ns, d], k [Nr, d], v [Nr, dv] with d % 16 == 0 expected")
        if nr == 0:
            raise ValueError("attention_core: softmax over zero keys")
        dv = v.size(1)
        dev = q.device
        L = _lib.lib()
        st = current_stream_ptr(dev)
        nsp, nrp = (ns + 31) // 32 * 32, _ceil_keys(nr)     # (32: whole stages of the dW-shaped kernels)
        bq = max(16, min(_ceil16(block_rows), nsp))
        # softmax over the keys is invariant to adding one vector c to every key (every score of a
        # row shifts by q.c): the whole computation runs on keys centred at their mean.  Same function,
        # same gradients (rows of dS sum to zero), but the cancellation-heavy sums of the backward -
        # dQ = dS K with sum_j dS_ij = 0 - no longer carry the keys' common component, which after a
        # ReLU encoder is most of their magnitude
        k = k - k[:nr].mean(dim=0, keepdim=True)
        qp, kp, vp = _pad_rows(q, nsp), _pad_rows(k, nrp), _pad_rows(v, nrp)
        # the images the backward needs (rows of K^T for dQ = dS K, rows of V for dP = dO V^T) come out of the same two
        # preparation launches
        need_bwd = any(ctx.needs_input_grad[:3])
        kmax, kimg, ktimg, ktmax = _prep(L, kp, need_bwd, st)      # rows of K over d (+ rows of K^T over the keys)
        vmax, vimg, vtimg, vtmax = _prep(L, vp, True, st)          # rows of V over dv, rows of V^T over the keys
        qmax = _rowabsmax(L, qp, st)
        ones = torch.ones(bq, dtype=torch.float32, device=dev)     # softmax weights are <= 1
        o = torch.empty((nsp, dv), dtype=torch.float32, device=dev)
        lse = torch.empty(nsp, dtype=torch.float32, device=dev)
        flash = FLASH and d == FLASH_D and dv == FLASH_D and nrp % 32 == 0
        s = None if flash else torch.empty((bq, nrp), dtype=torch.float32, device=dev)
        ws_o = None if flash else _splitk_ws(L, bq, nrp, dv, dev)
        if flash:
            kuns = torch.empty_like(kmax)
            _lib.check(L.dc_attn_flash_prep(vtimg.data_ptr(), dv, nrp, kmax.data_ptr(), kuns.data_ptr(), st),
                       "dc_attn_flash_prep")
            _lib.check(L.dc_attn_flash_fwd(qp.data_ptr(), d, qmax.data_ptr(), kimg.data_ptr(), kuns.data_ptr(),
                                           vtimg.data_ptr(), vtmax.data_ptr(), nsp, nr, nrp, d, o.data_ptr(), dv,
                                           lse.data_ptr(), st), "dc_attn_flash_fwd")
        for r0 in (() if flash else range(0, nsp, bq)):
            rows = min(bq, nsp - r0)
            _gemm(L, qp[r0:].data_ptr(), d, rows, d, kimg, nrp, s.data_ptr(), nrp, qmax[r0:].data_ptr(), kmax, st)
            _lib.check(L.dc_attn_softmax_rows(s.data_ptr(), nrp, rows, nr, nrp, lse[r0:].data_ptr(), st),
                       "dc_attn_softmax_rows")
            _gemm(L, s.data_ptr(), nrp, rows, nrp, vtimg, dv, o[r0:].data_ptr(), dv, ones.data_ptr(), vtmax, st, ws_o)
        if need_bwd:
            ctx.save_for_backward(qp, kp, vp, o, lse, kimg, kmax, qmax, vimg, vmax, ktimg, ktmax)
        ctx.dims = (ns, nr, d, dv, bq)
        return o[:ns]

    @staticmethod
    def backward(ctx, go: torch.Tensor):
        qp, kp, vp, o, lse, kimg, kmax, qmax, vimg, vmax, ktimg, ktmax = ctx.saved_tensors
        ns, nr, d, dv, bq = ctx.dims
        dev = qp.device
        L = _lib.lib()
        st = current_stream_ptr(dev)
        nsp, nrp = qp.size(0), kp.size(0)
        gop = _pad_rows(go.contiguous(), nsp)
        gomax = _rowabsmax(L, gop, st)
        if (FLASH_BWD and d == FLASH_D and dv == FLASH_D and nrp % 32 == 0
                and 8 * nsp * nrp <= _flash_bwd_budget(dev)):
            return _backward_flash(L, st, dev, qp, kp, gop, o, lse, kimg, kmax, qmax, gomax, vimg, vmax, ktimg, ktmax,
                                   ns, nr, d, dv)
        ones = torch.ones(bq, dtype=torch.float32, device=dev)
        gq = torch.empty((nsp, d), dtype=torch.float32, device=dev)
        gk = torch.empty((nrp, d), dtype=torch.float32, device=dev)
        gv = torch.empty((nrp, dv), dtype=torch.float32, device=dev)
        p = torch.empty((bq, nrp), dtype=torch.float32, device=dev)
        ds = torch.empty((bq, nrp), dtype=torch.float32, device=dev)
        dsmax = torch.empty(bq, dtype=torch.float32, device=dev)
        nb_k = L.dc_tag_linear_bwd_dw_workspace_bytes(bq, d, nrp, 1)
        nb_v = L.dc_tag_linear_bwd_dw_workspace_bytes(bq, dv, nrp, 1)
        scratch = torch.empty(max(nb_k, nb_v), dtype=torch.uint8, device=dev)
        ws_q = _splitk_ws(L, bq, nrp, d, dev)
        for r0 in range(0, nsp, bq):
            rows = min(bq, nsp - r0)
            acc = int(r0 > 0)
            # recompute the block's weights: scores + exp(s - lse) in the GEMM's epilogue
            _lib.check(L.dc_tag_linear_fwd_h2p_exp(qp[r0:].data_ptr(), d, kimg.data_ptr(), p.data_ptr(), nrp, rows,
                                                   d, nrp, qmax[r0:].data_ptr(), kmax.data_ptr(),
                                                   lse[r0:].data_ptr(), nr, st), "dc_tag_linear_fwd_h2p_exp")
            # dP = dO V^T, then dS = P * (dP - delta) in place
            _gemm(L, gop[r0:].data_ptr(), dv, rows, dv, vimg, nrp, ds.data_ptr(), nrp, gomax[r0:].data_ptr(), vmax, st)
            # delta formed inside the row kernel from the SAME P and dP, relative to the actual sum of the
            # recomputed weights (row sums of dS are then zero to rounding)
            _lib.check(L.dc_attn_ds_rows(p.data_ptr(), ds.data_ptr(), nrp, rows, nrp, None, dsmax.data_ptr(), st),
                       "dc_attn_ds_rows")
            # dQ_b = dS K
            _gemm(L, ds.data_ptr(), nrp, rows, nrp, ktimg, d, gq[r0:].data_ptr(), d, dsmax.data_ptr(), ktmax, st, ws_q)
            # dK += dS^T Q_b ; dV += P^T dO_b   (contraction over the block's rows: dW-shaped)
            for g_t, g_max, x_t, x_ld, x_max, out_t, fi, nb in (
                    (ds, dsmax, qp[r0:], d, qmax[r0:], gk, d, nb_k),
                    (p, ones, gop[r0:], dv, gomax[r0:], gv, dv, nb_v)):
                _lib.check(L.dc_tag_linear_bwd_dw_h2(
                    g_t.data_ptr(), nrp, None, nrp, _ptr_array([x_t]), _i64_array([x_ld]), 1,
                    _ptr_array([out_t]), 1, fi, None, acc, scratch.data_ptr(), nb, rows, fi, nrp,
                    g_max.data_ptr(), x_max.data_ptr(), st), "dc_tag_linear_bwd_dw_h2")
        return gq[:ns], gk[:nr], gv[:nr], None


def _backward_flash(L, st, dev, qp, kp, gop, o, lse, kimg, kmax, qmax, gomax, vimg, vmax, ktimg, ktmax, ns, nr, d, dv):
    """Backward with P and dS of ALL rows from ``dc_attn_flash_ds``, then dQ = dS K, dK = dS^T Q, dV = P^T dO as three
    launches over all rows.  Two sweeps per 128-query tile (delta from the same recomputed P and dP that form dS), or
    ONE (``FLASH_BWD_SINGLE``): dS' with delta = rowsum(dO * O) + the per-row difference eps to the consistent delta,
    which dK and dQ take into account at load time (dS' - eps o P)."""
    nsp, nrp = qp.size(0), kp.size(0)
    single = FLASH_BWD_SINGLE and nsp % 32 == 0 and nrp % 128 == 0
    kuns, vuns = torch.empty_like(kmax), torch.empty_like(vmax)
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, kmax.data_ptr(), kuns.data_ptr(), st), "dc_attn_flash_prep")
    _lib.check(L.dc_attn_flash_prep(None, 0, nrp, vmax.data_ptr(), vuns.data_ptr(), st), "dc_attn_flash_prep")
    p = torch.empty((nsp, nrp), dtype=torch.float32, device=dev)
    ds = torch.empty((nsp, nrp), dtype=torch.float32, device=dev)
    dsmax = torch.empty(nsp, dtype=torch.float32, device=dev)
    delta = (gop * o).sum(dim=1) if single else None
    eps = torch.empty(nsp, dtype=torch.float32, device=dev) if single else None
    _lib.check(L.dc_attn_flash_ds(qp.data_ptr(), d, qmax.data_ptr(), gop.data_ptr(), dv, gomax.data_ptr(),
                                  kimg.data_ptr(), kuns.data_ptr(), vimg.data_ptr(), vuns.data_ptr(), lse.data_ptr(),
                                  nsp, nr, nrp, d, p.data_ptr(), ds.data_ptr(), nrp, dsmax.data_ptr(),
                                  delta.data_ptr() if single else None, eps.data_ptr() if single else None, st),
               "dc_attn_flash_ds")
    gq = torch.empty((nsp, d), dtype=torch.float32, device=dev)
    gk = torch.empty((nrp, d), dtype=torch.float32, device=dev)
    gv = torch.empty((nrp, dv), dtype=torch.float32, device=dev)
    if single:
        _lib.check(L.dc_tag_linear_fwd_h2p_corr(ds.data_ptr(), nrp, p.data_ptr(), eps.data_ptr(), ktimg.data_ptr(),
                                                gq.data_ptr(), d, nsp, nrp, d, dsmax.data_ptr(), ktmax.data_ptr(), st),
                   "dc_tag_linear_fwd_h2p_corr")
    else:
        _gemm(L, ds.data_ptr(), nrp, nsp, nrp, ktimg, d, gq.data_ptr(), d, dsmax.data_ptr(), ktmax, st,
              _splitk_ws(L, nsp, nrp, d, dev))
    ones = torch.ones(nsp, dtype=torch.float32, device=dev)        # softmax weights are <= 1
    nb = max(L.dc_tag_linear_bwd_dw_workspace_bytes(nsp, d, nrp, 1), L.dc_tag_linear_bwd_dw_workspace_bytes(nsp, dv, nrp, 1))
    scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    for g_t, g_max, x_t, x_ld, x_max, out_t, fi in ((ds, dsmax, qp, d, qmax, gk, d), (p, ones, gop, dv, gomax, gv, dv)):
        if single and g_t is ds:
            _lib.check(L.dc_tag_linear_bwd_dw_h2_corr(
                ds.data_ptr(), nrp, p.data_ptr(), eps.data_ptr(), _ptr_array([x_t]), _i64_array([x_ld]), 1,
                _ptr_array([out_t]), 1, fi, 0, scratch.data_ptr(), nb, nsp, fi, nrp, g_max.data_ptr(), x_max.data_ptr(), st),
                "dc_tag_linear_bwd_dw_h2_corr")
            continue
        _lib.check(L.dc_tag_linear_bwd_dw_h2(
            g_t.data_ptr(), nrp, None, nrp, _ptr_array([x_t]), _i64_array([x_ld]), 1, _ptr_array([out_t]), 1, fi, None,
            0, scratch.data_ptr(), nb, nsp, fi, nrp, g_max.data_ptr(), x_max.data_ptr(), st), "dc_tag_linear_bwd_dw_h2")
    return gq[:ns], gk[:nr], gv[:nr], None


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                   block_rows: int = BLOCK_ROWS) -> torch.Tensor:
    """``softmax(q @ k.T, dim=-1) @ v`` without the ``[Ns, Nr]`` matrix (differentiable)."""
    return _AttnCoreFn.apply(q.contiguous(), k.contiguous(), v.contiguous(), int(block_rows))
