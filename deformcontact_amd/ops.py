"""Host-side operator layer: thin launches of the C ABI + autograd wiring.

Mirrors what PyG's ``TAGConv`` / ``GCNConv`` / ``GATConv`` ``forward`` do for the
reference (``/root/reference/models/model.py:71,77``), with every gather /
scatter step executed by ``libdeformcontact_hip.so`` on the current HIP stream.
PyTorch is used for device memory, autograd bookkeeping and (for now) the plain
dense GEMMs.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _lib
from .deferred import resolve
from .graph import GraphIndex, SortedAdjacency, _require_cuda, current_stream_ptr


def _rowmajor(t: torch.Tensor, what: str) -> int:
    """Return the leading dimension of a 2-D fp32 row-major (possibly column-sliced) view."""
    if t.dim() != 2 or t.dtype != torch.float32:
        raise ValueError(f"{what}: expected a 2-D float32 tensor, got {tuple(t.shape)} {t.dtype}")
    if t.size(1) > 1 and t.stride(1) != 1:
        raise ValueError(f"{what}: innermost dimension must be contiguous")
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def hop(adj: SortedAdjacency, x: torch.Tensor, out: Optional[torch.Tensor] = None,
        addend: Optional[torch.Tensor] = None, weighted: bool = True,
        rowmax: Optional[torch.Tensor] = None, rowmax_mode: int = 0) -> torch.Tensor:
    """``out[i] = addend[i] + sum_{p in seg(i)} w[p] * x[other[p]]`` (one launch).

    ``x`` / ``out`` / ``addend`` may be column slices of wider row-major buffers.  With ``rowmax``
    (float32 ``[N]``) the launch also records ``max |out[i, :]|`` there (``rowmax_mode`` bit 0:
    joined with ``max |x[i, :]|``, bit 1: joined with the value already stored)."""
    _require_cuda(x, "x")
    n, f = x.shape
    if adj.ptr.numel() != n + 1:
        raise ValueError(f"hop: x has {n} rows but the graph has {adj.ptr.numel() - 1} nodes")
    if out is None:
        out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    ldx, ldy = _rowmajor(x, "x"), _rowmajor(out, "out")
    if out.shape != x.shape:
        raise ValueError("hop: out shape mismatch")
    lda = 0
    if addend is not None:
        if addend.shape != x.shape:
            raise ValueError("hop: addend shape mismatch")
        lda = _rowmajor(addend, "addend")
    w = adj.w if weighted else None
    if rowmax is not None and (rowmax.dtype != torch.float32 or rowmax.numel() != n or not rowmax.is_contiguous()):
        raise ValueError("hop: rowmax must be a contiguous float32 [N] tensor")
    if adj.row_offset:
        # a row window of a merged adjacency: x / out / addend / rowmax hold only the window's rows
        if rowmax is not None:
            rc = _lib.lib().dc_spmm_f32_rowmax_window(
                adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
                x.data_ptr(), ldx, addend.data_ptr() if addend is not None else None, lda,
                out.data_ptr(), ldy, n, f, rowmax.data_ptr(), int(rowmax_mode), int(adj.row_offset),
                current_stream_ptr(x.device))
            _lib.check(rc, "dc_spmm_f32_rowmax_window")
            return out
        rc = _lib.lib().dc_spmm_f32_window(
            adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
            x.data_ptr(), ldx, addend.data_ptr() if addend is not None else None, lda,
            out.data_ptr(), ldy, n, f, int(adj.row_offset), current_stream_ptr(x.device))
        _lib.check(rc, "dc_spmm_f32_window")
        return out
    if rowmax is not None:
        rc = _lib.lib().dc_spmm_f32_rowmax(
            adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
            x.data_ptr(), ldx, addend.data_ptr() if addend is not None else None, lda,
            out.data_ptr(), ldy, n, f, rowmax.data_ptr(), int(rowmax_mode),
            current_stream_ptr(x.device))
        _lib.check(rc, "dc_spmm_f32_rowmax")
        return out
    rc = _lib.lib().dc_spmm_f32(
        adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
        x.data_ptr(), ldx, addend.data_ptr() if addend is not None else None, lda,
        out.data_ptr(), ldy, n, f, current_stream_ptr(x.device))
    _lib.check(rc, "dc_spmm_f32")
    return out


def weight_rowmax(ws) -> torch.Tensor:
    """``out[o] = max_s,f |W_s[o, f]|`` over the K+1 weight blocks of a TAGConv layer
    (``dc_tag_weight_rowmax``): the per-output-row scale of the fp16x2 dense block."""
    fo, fi = ws[0].shape
    out = torch.empty(fo, dtype=torch.float32, device=ws[0].device)
    rc = _lib.lib().dc_tag_weight_rowmax(_ptr_array(ws), len(ws), fo, fi, out.data_ptr(),
                                         current_stream_ptr(out.device))
    _lib.check(rc, "dc_tag_weight_rowmax")
    return out


def rowabsmax(x: torch.Tensor) -> torch.Tensor:
    """``out[i] = max |x[i, :]|`` of a row-major 2-D float32 view (``dc_rowabsmax_f32``)."""
    _require_cuda(x, "x")
    n, f = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    rc = _lib.lib().dc_rowabsmax_f32(x.data_ptr(), _rowmajor(x, "x"), n, f, out.data_ptr(),
                                     current_stream_ptr(x.device))
    _lib.check(rc, "dc_rowabsmax_f32")
    return out


def _rowmajor_any(t: torch.Tensor, what: str, dtypes) -> int:
    if t.dim() != 2 or t.dtype not in dtypes:
        raise ValueError(f"{what}: expected a 2-D tensor of {dtypes}, got {tuple(t.shape)} {t.dtype}")
    if t.size(1) > 1 and t.stride(1) != 1:
        raise ValueError(f"{what}: innermost dimension must be contiguous")
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def hop_bf16(adj: SortedAdjacency, x: torch.Tensor, out: Optional[torch.Tensor] = None,
             addend: Optional[torch.Tensor] = None, weighted: bool = True,
             out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """The hop over bf16-stored features with fp32 accumulation (``dc_spmm_bf16``; SURVEY.md 8(d)
    config 5).  ``x`` is bfloat16; ``out`` / ``addend`` are ``out_dtype`` (bfloat16: the fp32 sum
    is rounded once on store; float32: the fp32 sum itself)."""
    _require_cuda(x, "x")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("hop_bf16: out_dtype must be bfloat16 or float32")
    ldx = _rowmajor_any(x, "x", (torch.bfloat16,))
    n, f = x.shape
    if adj.ptr.numel() != n + 1:
        raise ValueError(f"hop_bf16: x has {n} rows but the graph has {adj.ptr.numel() - 1} nodes")
    if out is None:
        out = torch.empty((n, f), dtype=out_dtype, device=x.device)
    if out.shape != x.shape:
        raise ValueError("hop_bf16: out shape mismatch")
    ldy = _rowmajor_any(out, "out", (out_dtype,))
    lda = 0
    if addend is not None:
        if addend.shape != x.shape:
            raise ValueError("hop_bf16: addend shape mismatch")
        lda = _rowmajor_any(addend, "addend", (out_dtype,))
    w = adj.w if weighted else None
    rc = _lib.lib().dc_spmm_bf16(
        adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
        x.data_ptr(), ldx, addend.data_ptr() if addend is not None else None, lda,
        out.data_ptr(), ldy, n, f, 1 if out_dtype == torch.float32 else 0,
        current_stream_ptr(x.device))
    _lib.check(rc, "dc_spmm_bf16")
    return out


class _HopFn(torch.autograd.Function):
    """Differentiable single hop ``y = A x`` (``A`` = weighted sorted adjacency)."""

    @staticmethod
    def forward(ctx, g: GraphIndex, x: torch.Tensor, weighted: bool):
        ctx.g, ctx.weighted = g, weighted
        return hop(g.fwd, x.contiguous(), weighted=weighted)

    @staticmethod
    def backward(ctx, gy):
        return None, hop(ctx.g.bwd, gy.contiguous(), weighted=ctx.weighted), None


def propagate(g: GraphIndex, x: torch.Tensor, weighted: bool = True) -> torch.Tensor:
    """Autograd-aware hop (PyG ``propagate(edge_index, x=x, edge_weight=w)``)."""
    return _HopFn.apply(g, resolve(x), weighted)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _i64_array(vals):
    return (ctypes.c_int64 * len(vals))(*vals)


MAX_SEG = 4   # DC_MAX_SEG in include/deformcontact.h

#: Direct parameter-gradient mode (opt-in, ``dp.GradBucket(params, direct=True)``): the dW
#: slab-reduce kernel accumulates weight / bias gradients straight into the bucket's views instead
#: of returning them to autograd (which would run one elementwise ``add`` kernel per parameter).
#: Only parameters the bucket has marked are written that way (``requires_grad`` ones whose
#: ``.grad`` still is the bucket's view); every such write is reported to the bucket
#: (``GradBucket.note_direct_write``), which makes its consumers (``all_reduce_mean``,
#: ``FlatAdam.step``) wait for the writing stream.  Tensor hooks on those parameters do not fire
#: and ``torch.autograd.grad`` gets ``None`` for them - hence opt-in.  Setting this to False
#: disables the mode globally.
DIRECT_PARAM_GRAD = True


#: Run the dense blocks (forward, dX, dW) on the bf16 matrix cores with every fp32 operand split
#: exactly into hi + mid + lo bf16 terms and six MFMA products (``dc_tag_linear_*_split``):
#: fp32-accurate (measured error vs float64 equal to the fp32-MFMA / rocBLAS kernels,
#: profiles/r01/d_split_accuracy.txt) at up to 2.7x the fp32 MFMA peak.  ``DC_DENSE_SPLIT=0``
#: (or setting this to False) keeps everything on ``v_mfma_f32_32x32x2_f32``.
DENSE_SPLIT_BF16 = os.environ.get("DC_DENSE_SPLIT", "1") != "0"

#: bf16 MFMA products per tile in split mode: 6 = fp32-accurate (the C-ABI also takes 3 = ~4e-6 relative and
#: 1 = plain bf16 operands with fp32 accumulation; nothing in this package asks for them)
DENSE_PRODUCTS = 6


#: the short-reduction forward kernel for the first layers (``dc_tag_linear_fwd_narrow``: persistent, weights resident in
#: registers, no packing launch; bit-identical to the six-product split kernel); ``DC_NARROW_FWD=0``: ``dc_tag_pack_weights`` +
#: ``dc_tag_linear_fwd_split`` as up to round 5
NARROW_FWD = os.environ.get("DC_NARROW_FWD", "1") != "0"

#: fp16x2 mode of the wide (Fi % 16 == 0, unconcatenated) dense blocks: two power-of-two-scaled
#: fp16 planes per operand and THREE MFMA products instead of the six of the bf16 split - still
#: fp32-accurate (error below that of fp32 accumulation), half the matrix work.  The row maxima
#: the scaling needs come out of the hop launches.  ``DC_DENSE_F16X2=0`` keeps the bf16x3 split.
DENSE_F16X2 = os.environ.get("DC_DENSE_F16X2", "1") != "0"


#: K chained hops of a batch with a known layout as ONE launch with every graph's slice resident in LDS
#: (``dc_hop_chain_f32``, bit-identical to the K single hops); ``DC_HOP_CHAIN=0``: hop by hop.
HOP_CHAIN = os.environ.get("DC_HOP_CHAIN", "1") != "0"
_CHAIN_MAX_NODES = None


def hop_chain_eligible(g, adj: SortedAdjacency, slab: torch.Tensor, f: int, k: int) -> bool:
    """Can ``dc_hop_chain_f32`` run this chain?  Needs the batch layout (``graph_index(..., segments=)``), graphs
    of at most ``dc_hop_chain_max_nodes()`` nodes (4,096: 32-column slices up to 1,024 nodes, 16 up to 2,048, 8 beyond),
    F % 32 == 0 and 16-byte aligned slab rows."""
    global _CHAIN_MAX_NODES
    seg = getattr(g, "_layout", None)
    if not HOP_CHAIN or seg is None or k < 1 or f % 32 != 0 or adj.row_offset:
        return False
    if _CHAIN_MAX_NODES is None:
        _CHAIN_MAX_NODES = int(_lib.lib().dc_hop_chain_max_nodes())
    return (0 < g._seg_max_nodes <= _CHAIN_MAX_NODES and slab.dtype == torch.float32 and slab.dim() == 2
            and slab.stride(1) == 1 and slab.stride(0) % 4 == 0 and slab.data_ptr() % 16 == 0
            and adj.ptr.numel() == slab.size(0) + 1)


#: re-form gcn_norm weights from an LDS-resident degree table inside ``dc_hop_chain_f32`` (no vector-memory loads in its
#: hop loop) instead of loading ``w``; same bits.  False (tests): always load them.  (Round 4 kept graphs of up to
#: 512 nodes off this form because of a rare run-to-run difference; round 5 traced that to other kernels' workgroups sharing
#: the compute unit's LDS with a small chain workgroup and removed the condition itself - every chain workgroup now owns
#: the whole LDS, ``dc_hopchain.hip: kChainLdsRequest`` - so the size rule is gone.)
HOP_CHAIN_GCN = True


def hop_chain(g, adj: SortedAdjacency, slab: torch.Tensor, f: int, k: int, weighted: bool = True,
              rowmax: Optional[torch.Tensor] = None, rowmax_mode: int = 0, src_block: int = 0, direction: int = 1) -> None:
    """``dc_hop_chain_f32``: blocks ``src_block + direction .. src_block + k * direction`` of ``slab`` from block
    ``src_block``, one launch (see ``hop_chain_eligible``)."""
    nptr, nseg = g._layout
    w = adj.w if weighted else None
    # the adjacency's weights are gcn_norm's (graph.GraphIndex builds nothing else): tell the kernel so
    deg = g.fwd.ptr if (HOP_CHAIN_GCN and w is not None and g.normalize and not g.self_loops) else None
    rc = _lib.lib().dc_hop_chain_f32(
        adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr() if w is not None else None,
        deg.data_ptr() if deg is not None else None, adj.other.numel(),
        nptr, nseg, slab.data_ptr(), slab.stride(0), slab.size(0), f, k, int(src_block), int(direction),
        rowmax.data_ptr() if rowmax is not None else None, int(rowmax_mode), current_stream_ptr(slab.device))
    _lib.check(rc, "dc_hop_chain_f32")


#: diagnostic tap (tools/exp/dp_flake2.py): ``DEBUG_TAP(name, tensor)`` is called with intermediate tensors of
#: ``_TagConvFn.backward`` when set; None in production
DEBUG_TAP = None
#: diagnostic hook (tools/exp/chain_hunt.py): called as ``DEBUG_CHAIN(g, adj, slab, f, k, transposed)`` right after a
#: ``dc_hop_chain_f32`` launch of ``chained_hops`` (``DEBUG_CHAIN_PRE``: right before it); None in production
DEBUG_CHAIN = None
DEBUG_CHAIN_PRE = None


def chained_hops(g: GraphIndex, slab: torch.Tensor, f: int, k: int, backward: bool,
                 rowmax: Optional[torch.Tensor] = None, transposed: bool = False,
                 rowmax_has_block0: bool = False, rowmax_zeroed: bool = False) -> None:
    """In place on ``slab`` ([N, ld], K+1 column blocks of width ``f``).  Forward: block j+1 =
    A block j (j = 0..k-1).  Backward: block j-1 += A^T block j (j = k..1).  Forward with
    ``rowmax`` ([N]): also ``rowmax[i] = max_j max |block j [i, :]|`` (needs k >= 1;
    ``rowmax_has_block0``: it already holds the maxima of block 0).  ``transposed`` (forward
    direction only): block j+1 = A^T block j."""
    if k == 0:
        return
    adj = g.bwd if (backward or transposed) else g.fwd
    blocks = [slab[:, j * f:(j + 1) * f] for j in range(k + 1)]
    if backward:
        for j in range(k, 0, -1):
            hop(adj, blocks[j], out=blocks[j - 1], addend=blocks[j - 1], weighted=g.normalize)
    elif hop_chain_eligible(g, adj, slab, f, k):
        # mode: 1 = fresh maxima over blocks 0..K (the entry clears the buffer first), 2 = joined with block 0's, which
        # the buffer already holds, 3 = over blocks 0..K joined with a buffer the caller has cleared (``rowmax_zeroed``)
        if DEBUG_CHAIN_PRE is not None:
            DEBUG_CHAIN_PRE(g, adj, slab, f, k, transposed)
        hop_chain(g, adj, slab, f, k, weighted=g.normalize, rowmax=rowmax,
                  rowmax_mode=2 if rowmax_has_block0 else (3 if rowmax_zeroed else 1))
        if DEBUG_CHAIN is not None:
            DEBUG_CHAIN(g, adj, slab, f, k, transposed)
    else:
        for j in range(k):
            hop(adj, blocks[j], out=blocks[j + 1], weighted=g.normalize, rowmax=rowmax,
                rowmax_mode=1 if (j == 0 and not rowmax_has_block0) else 2)


#: Cache the hop slab ``[x | A x | ... | A^K x]`` (+ row maxima) of TAGConv layers whose input
#: needs no gradient - the FIRST layer of each branch (``models/model.py:71,77`` with the raw
#: ``graph.x``): it is a parameter-free function of (x, edge_index), so it is computed once per
#: batch (by the first forward, or ahead of time by ``precompute_input_hops`` on the loader's
#: stream) instead of once per step.  Keyed on the input tensor's address + version and stored on
#: the ``GraphIndex`` (i.e. dropped with the topology); under stream capture the same rule as for
#: the adjacency applies (``graph.graph_index``).  ``DC_HOP_CACHE=0`` disables.
HOP_CACHE = os.environ.get("DC_HOP_CACHE", "1") != "0"
_HOP_CACHE_ENTRIES = 2


def _hop_cache_key(x: torch.Tensor, k: int, wpad: int, want_rowmax: bool):
    return (x.data_ptr(), tuple(x.shape), tuple(x.stride()), int(k), int(wpad), bool(want_rowmax))


def _hop_cache_get(g, key, x, dev):
    """(slab, rowmax) cached for this input tensor in its CURRENT state, or None."""
    cache = getattr(g, "_hop_cache", None)
    if not cache or key not in cache:
        return None
    from .graph import capture_id
    cid = capture_id(dev)
    slab, rowmax, ecid, _ref, version = cache[key]
    if version == x._version and (ecid == cid or (cid != 0 and g._static_ok)):
        return slab, rowmax
    return None


def _hop_cache_put(g, key, x, slab, rowmax, dev):
    from .graph import capture_id
    cache = g.__dict__.setdefault("_hop_cache", {})
    if key not in cache:
        while len(cache) >= _HOP_CACHE_ENTRIES:
            cache.pop(next(iter(cache)))
    # x kept alive: its address is part of the key
    cache[key] = (slab, rowmax, capture_id(dev), x, x._version)


#: pack + first hop of a narrow layer's input in one launch (``dc_spmm_f32_pack``); ``DC_FUSED_PACK=0``: two launches
FUSED_PACK = os.environ.get("DC_FUSED_PACK", "1") != "0"


def _build_input_slab(g: GraphIndex, x: torch.Tensor, k: int, want_rowmax: bool, into=None):
    """Pack ``x`` into block 0 of a ``[N, wpad]`` slab (fresh, or the buffers ``into`` = (slab,
    rowmax) of an earlier call) and run the K hops (+ row maxima)."""
    n, fi = x.shape
    concat, width, wpad = tag_slab_geometry(fi, k)
    dev = x.device
    slab = into[0] if into is not None else _alloc_slab(n, wpad, dev)
    xin = x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous()
    adj = g.fwd if (g is not None and k >= 1) else None
    if (FUSED_PACK and adj is not None and not want_rowmax and fi <= 32 and not adj.row_offset and g.normalize
            and adj.ptr.numel() == n + 1):
        # the packing pass and the first hop in one launch (the chain of small dependent kernels a new batch starts with)
        _lib.check(_lib.lib().dc_spmm_f32_pack(adj.ptr.data_ptr(), adj.other.data_ptr(), adj.w.data_ptr(),
                                               xin.data_ptr(), xin.stride(0), slab.data_ptr(), slab.stride(0), n, fi,
                                               width, wpad, current_stream_ptr(dev)), "dc_spmm_f32_pack")
        for j in range(1, k):
            hop(adj, slab[:, j * fi:(j + 1) * fi], out=slab[:, (j + 1) * fi:(j + 2) * fi], weighted=True)
        return slab, None
    _lib.check(_lib.lib().dc_tag_pack_input(xin.data_ptr(), xin.stride(0), slab.data_ptr(), slab.stride(0), n,
                                            fi, width, wpad, current_stream_ptr(dev)),
               "dc_tag_pack_input")
    rowmax = None
    if want_rowmax:
        rowmax = into[1] if into is not None else torch.empty(n, dtype=torch.float32, device=dev)
    chained_hops(g, slab, fi, k, backward=False, rowmax=rowmax)
    return slab, rowmax


#: K = 0 layers (``dense_linear``: the ``lin`` of GCNConv / GATConv, the attention heads' Linear, the decoder) on the
#: fp16x2 kernels too: no hop records their row maxima, so a ``dc_rowabsmax_f32`` pass over the input is added (8 us for
#: [32768, 256]) and three MFMA products replace six.  From 128 input columns on, outputs a multiple of 16 wide
#: (the 256 -> 3 output layer stays where it was).


def _tag_uses_h2(fi: int, k: int, fo: Optional[int] = None) -> bool:
    concat, _, wpad = tag_slab_geometry(fi, k)
    ok = (DENSE_F16X2 and DENSE_SPLIT_BF16 and DENSE_PRODUCTS == 6 and not concat and fi % 16 == 0 and wpad % 4 == 0)
    if k >= 1:
        return ok
    return ok and fo is not None and fi >= 128 and fi % 32 == 0 and fo % 16 == 0 and fo >= 64


def precompute_input_hops(g: GraphIndex, x: torch.Tensor, k: int = 3) -> None:
    """Compute and cache, on the current stream, the hop slab a ``TAGConv(in, out, K=k)`` layer
    will need for the no-grad input ``x`` over topology ``g`` (what ``loaders.PrefetchLoader`` does
    for the next batch while the current one trains).

    (Until round 5 a ``refresh=True`` mode recomputed INTO the cached buffers of refilled static inputs.  Nothing used it,
    and it was unsafe: re-filing an eagerly allocated slab under the id of the capture that refreshed it made the next EAGER
    lookup miss, replace the entry and free a buffer whose address a captured graph still wrote to - a memory access
    fault at batch 32 when round 5 tried it (``profiles/r05/j_refresh_mode_memory_fault.txt``).  Removed; refilled static inputs rebuild their
    slabs inside the captured step, as ``bench.py`` does.)"""
    _require_cuda(x, "x")
    if not HOP_CACHE or k < 1 or x.dtype != torch.float32 or x.dim() != 2:
        return
    fi = x.size(1)
    want = _tag_uses_h2(fi, k)
    key = _hop_cache_key(x, k, tag_slab_geometry(fi, k)[2], want)
    if _hop_cache_get(g, key, x, x.device) is None:
        slab, rowmax = _build_input_slab(g, x, k, want)
        _hop_cache_put(g, key, x, slab, rowmax, x.device)


def _grad_sink(p):
    """The ``dp.GradBucket`` that owns ``p.grad`` in direct mode, or None."""
    ref = getattr(p, "_dc_grad_sink", None)
    bucket = ref() if ref is not None else None
    if bucket is None or not p.requires_grad:
        return None
    g = p.grad
    if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device
            or g.shape != p.shape or not bucket.owns(p, g)):
        return None
    return bucket


def tag_slab_geometry(fi: int, k: int):
    """(concat, width, padded width) of the ``[N, (K+1)*Fi]`` hop slab of a TAGConv layer."""
    concat = (fi * (k + 1) <= 128) or (fi % 16 != 0)
    width = (k + 1) * fi
    wpad = (width + 15) // 16 * 16 if concat else width
    return concat, width, wpad


_SLAB_TAG = "_dc_hop_slab"

#: floats of padding behind every row of a hop slab whose row is a multiple of 1 KiB: with the natural
#: leading dimension (1024 floats = 4 KiB for the hidden-256 layers) the rows of a tile / the neighbour
#: rows of a hop sit a power of two apart and alias in the memory channels - measured on MI355X
#: (tools/exp/ld_pad.py, operands from beyond the Infinity Cache): F = 256 hop 20.7 -> 18.3 us, wide
#: forward block 92.6 -> 82.3 us with 64 floats (256 B) of padding; +16 / +80 floats (rows no longer
#: 128-byte aligned) give nothing.
SLAB_PAD = 64


def _alloc_slab(n: int, wpad: int, dev, tag=None) -> torch.Tensor:
    """``[n, wpad]`` view of a fresh row-major buffer whose leading dimension is padded off the power of
    two (``SLAB_PAD``); ``tag`` marks the BASE as a slab this library owns (``_as_slab_block0``)."""
    pad = SLAB_PAD if (SLAB_PAD > 0 and (wpad * 4) % 1024 == 0) else 0
    base = torch.empty((n, wpad + pad), dtype=torch.float32, device=dev)
    if tag is not None:
        setattr(base, _SLAB_TAG, tag)
    return base[:, :wpad] if pad else base


def _as_slab_block0(x: torch.Tensor, n: int, fi: int, wpad: int):
    """If ``x`` is column block 0 of a ``[n, wpad]`` hop slab THIS LIBRARY allocated for it (the
    previous layer's forward wrote its output there and tagged the buffer), return that slab (a
    ``[n, wpad]`` view of the possibly row-padded buffer); else None.  Shape and stride alone are not
    enough: a caller's own ``feat[:, :fi]`` view of a wider tensor looks the same, and the hops would
    overwrite its other columns."""
    base = x._base
    if (base is not None and getattr(base, _SLAB_TAG, None) == (n, fi, wpad)
            and base.dim() == 2 and base.size(0) == n and base.size(1) >= wpad
            and base.is_contiguous() and x.stride() == (base.size(1), 1) and x.shape == (n, fi)
            and x.data_ptr() == base.data_ptr() and base.dtype == torch.float32):
        return base[:, :wpad] if base.size(1) > wpad else base
    return None


def _h2_weight_prep(L, ws, k: int, fo: int, fi: int, want_t: bool, dev, st, zero: Optional[torch.Tensor] = None):
    """``dc_tag_weight_prep(_zero)``: row maxima and scaled fp16x2 image of the layer's weights over the concatenated
    reduction, the same for the transposed weights (``want_t``: the forward-shaped dX block), and - on the side -
    ``zero`` cleared.  -> (wmax, wimg, wt, wt_rowmax)."""
    width = (k + 1) * fi
    wmax = torch.empty(fo, dtype=torch.float32, device=dev)
    wimg = torch.empty((fo, width), dtype=torch.float32, device=dev)     # 4 B / element
    wt = wt_rowmax = None
    if want_t:
        wt = torch.empty((fi, (k + 1) * fo), dtype=torch.float32, device=dev)
        wt_rowmax = torch.empty(fi, dtype=torch.float32, device=dev)
    _lib.check(L.dc_tag_weight_prep_zero(_ptr_array(ws), k + 1, fo, fi, wmax.data_ptr(), wimg.data_ptr(),
                                         wt.data_ptr() if wt is not None else None,
                                         wt_rowmax.data_ptr() if wt is not None else None,
                                         zero.data_ptr() if zero is not None else None,
                                         zero.numel() if zero is not None else 0, st), "dc_tag_weight_prep")
    return wmax, wimg, wt, wt_rowmax


#: In the backward of a wide layer dW needs the masked gradient and the forward's slab only - not the transposed hop chain,
#: not dX - so its place in the sequence mask -> chain -> dX is free.  With both encoder branches running the same sequence
#: on two streams the two ``k_dw_h2w`` launches run side by side - two launches of one 512-thread workgroup per CU each - and
#: take 151 us together where they take 49 + 66 us one after the other (``profiles/r05/w_step_timeline_headline_*.txt``).
#: Staggered: dW between chain and dX everywhere EXCEPT on the streams listed here (``graphnet.ContactEncoder`` registers the
#: side stream its rigid branch runs on), where it stays behind dX.  Measured, three boxes (``profiles/r05/x_dw_first.txt``):
#: 0.617 ms per step against 0.627 with both branches in the old order (dW in front of the chain on the soft branch: between
#: -7 and +5 us depending on the box; any other combination: slower).  Same kernels, same operands: same bits.
DW_LAST_STREAMS: set = set()
#: where dW goes in a wide layer's backward, by class of the current stream ("listed" = in DW_LAST_STREAMS): "first" (in front
#: of the transposed chain), "mid" (between chain and dX) or "last" (behind dX: the order of rounds 1 - 4 on both streams)
DW_POSITION = {"unlisted": "mid", "listed": "last"}


def _dw_position(dev) -> str:
    return DW_POSITION["listed" if current_stream_ptr(dev) in DW_LAST_STREAMS else "unlisted"]


class _TagConvFn(torch.autograd.Function):
    """Whole TAGConv layer (+ optional fused ReLU): K hops into one ``[N, (K+1)*Fi]`` slab, then
    ONE fp32-MFMA kernel for ``act(x W_0^T + sum_k (A^k x) W_k^T + b)`` - PyG ``tag_conv.py``
    forward followed by ``F.relu`` (``models/model.py:71,77``).  Backward: one dW kernel
    (+ bias grad), one dX kernel writing the per-hop gradient slab, K transposed hops.

    Narrow layers (Fi = 21 / 25 of the input encodings) run the dense block as ONE segment
    over the concatenated slab (K-dim (K+1)*Fi = 84 / 100: aligned float4 loads, one
    128-wide dW tile); wide layers pass the K+1 column blocks as separate segments."""

    @staticmethod
    def forward(ctx, g: GraphIndex, x: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
                next_geom, *weights):
        n, fi = x.shape
        k = len(weights) - 1
        fo = weights[0].size(0)
        if k + 1 > MAX_SEG:
            raise NotImplementedError(f"TAGConv K={k} > {MAX_SEG - 1} is not supported by the fused dense block")
        dev = x.device
        concat, width, wpad = tag_slab_geometry(fi, k)
        # narrow layers: one K segment over the whole slab, zero-padded to a multiple of 16 so
        # the lean MFMA path (aligned float4 loads, no K tail) applies (84 -> 96, 100 -> 112)
        slab = _as_slab_block0(x, n, fi, wpad)
        if (slab is None and k == 0 and wpad == fi and x.is_contiguous() and x.data_ptr() % 16 == 0):
            slab = x                                 # no hops: the input itself is the (1-block) slab
        L = _lib.lib()
        st = current_stream_ptr(dev)
        six = next_geom is SIX_PRODUCTS
        if six:
            next_geom = None
        h2 = _tag_uses_h2(fi, k, None if six else fo)
        rowmax = None
        prepped = None
        if slab is None:
            # the layer's own input: pack + K hops, or the cached slab when x needs no gradient
            key = None
            if HOP_CACHE and g is not None and k >= 1 and not ctx.needs_input_grad[1]:
                key = _hop_cache_key(x, k, wpad, h2)
                hit = _hop_cache_get(g, key, x, dev)
                if hit is not None:
                    slab, rowmax = hit
            if slab is None:
                slab, rowmax = _build_input_slab(g, x, k, h2)
                if key is not None:
                    _hop_cache_put(g, key, x, slab, rowmax, dev)
        else:
            rowmax = torch.empty(n, dtype=torch.float32, device=dev) if h2 else None
            zeroed = False
            if h2 and g is not None and hop_chain_eligible(g, g.fwd, slab, fi, k):
                # the chain launch joins its row maxima into `rowmax` with atomics: the weight preparation - one launch
                # anyway, independent of the slab - clears it on the side (a memset node of its own: ~5 us per chain)
                prepped = _h2_weight_prep(L, [w.contiguous() for w in weights], k, fo, fi,
                                          ctx.needs_input_grad[1] and fo % 16 == 0, dev, st, zero=rowmax)
                zeroed = True
            chained_hops(g, slab, fi, k, backward=False, rowmax=rowmax, rowmax_zeroed=zeroed)
        if h2 and k == 0:
            # no hop has recorded the rows' maxima: one pass over the input
            rowmax = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(L.dc_rowabsmax_f32(slab.data_ptr(), slab.stride(0), n, fi, rowmax.data_ptr(), st), "dc_rowabsmax_f32")
        blocks = [slab[:, j * fi:(j + 1) * fi] for j in range(k + 1)]
        narrow = (concat and NARROW_FWD and DENSE_SPLIT_BF16 and DENSE_PRODUCTS == 6
                  and bool(L.dc_tag_linear_fwd_narrow_ok(fi, k + 1, wpad, fo)))
        if concat and narrow:
            # the short-reduction kernel gathers its weight fragments from the lins[k].weight matrices themselves
            ws = [w.contiguous() for w in weights]
            xs, ldxs, fi_eff = [slab], [slab.stride(0)], wpad
        elif concat:
            wc = [w.contiguous() for w in weights]
            wcat = torch.empty((fo, wpad), dtype=torch.float32, device=dev)
            _lib.check(L.dc_tag_pack_weights(_ptr_array(wc), k + 1, wcat.data_ptr(), fo, fi, wpad,
                                             st), "dc_tag_pack_weights")
            ws = [wcat]                                              # [Fo, wpad]
            xs, ldxs, fi_eff = [slab], [slab.stride(0)], wpad
        else:
            ws = [w.contiguous() for w in weights]
            xs, ldxs, fi_eff = blocks, [slab.stride(0)] * (k + 1), fi
        if isinstance(next_geom, OutInto):
            # the output goes into rows of a buffer the caller owns (a part's rows of block 0 of a MERGED
            # hop slab: both encoder branches feed one grouped layer)
            out = next_geom.view
            if out.shape != (n, fo) or out.stride(1) != 1 or out.dtype != torch.float32 or out.device != dev:
                raise ValueError("tag_conv: out_into view has the wrong shape / layout")
        elif next_geom is not None:
            # the output IS column block 0 of the next TAGConv layer's hop slab (no copy there)
            next_width, next_wpad = next_geom
            nxt = _alloc_slab(n, next_wpad, dev, tag=(n, fo, next_wpad))   # recognised by _as_slab_block0
            if next_wpad > next_width:
                nxt[:, next_width:].zero_()          # K padding of a narrow next layer
            out = nxt[:, :fo]
        else:
            out = torch.empty((n, fo), dtype=torch.float32, device=dev)
        ldo = out.stride(0)
        b = bias.contiguous() if bias is not None else None
        args = (_ptr_array(xs), _i64_array(ldxs), _ptr_array(ws), len(xs),
                b.data_ptr() if b is not None else None, int(relu), out.data_ptr(), ldo, n, fi_eff, fo)
        wmax = wt = wt_rowmax = None
        if h2:
            # one launch: the weights scaled and split into their fp16 planes over the concatenated
            # reduction (the dense block runs as ONE segment over the whole slab and pulls them into
            # LDS by DMA), their row maxima and, when the input needs a gradient, the same for the
            # transposed weights (forward-shaped dX block)
            if prepped is None:
                prepped = _h2_weight_prep(L, ws, k, fo, fi, ctx.needs_input_grad[1] and fo % 16 == 0, dev, st)
            wmax, wimg, wt, wt_rowmax = prepped
            rc = L.dc_tag_linear_fwd_h2p(slab.data_ptr(), slab.stride(0), wimg.data_ptr(),
                                         b.data_ptr() if b is not None else None, int(relu),
                                         out.data_ptr(), ldo, n, width, fo,
                                         rowmax.data_ptr(), wmax.data_ptr(), None, 0, st)
        elif narrow:
            rc = L.dc_tag_linear_fwd_narrow(slab.data_ptr(), slab.stride(0), _ptr_array(ws), k + 1, fi,
                                            b.data_ptr() if b is not None else None, int(relu), out.data_ptr(), ldo, n,
                                            wpad, fo, st)
        elif DENSE_SPLIT_BF16:
            rc = L.dc_tag_linear_fwd_split(*args, DENSE_PRODUCTS, st)
        else:
            rc = L.dc_tag_linear_fwd(*args, st)
        _lib.check(rc, "dc_tag_linear_fwd")
        ctx.g, ctx.k, ctx.fi, ctx.fo, ctx.has_bias, ctx.relu, ctx.concat = \
            g, k, fi, fo, bias is not None, relu, concat
        ctx.narrow = narrow
        ctx.params, ctx.bias_param = weights, bias       # the Parameter objects themselves
        ctx.h2 = h2
        ctx.save_for_backward(slab, out if relu else None, rowmax, wt, wt_rowmax, *ws)
        return out

    @staticmethod
    def backward(ctx, gout):
        slab, out, xrowmax, wt, wt_rowmax, *ws = ctx.saved_tensors
        g, k, fi, fo, concat = ctx.g, ctx.k, ctx.fi, ctx.fo, ctx.concat
        L = _lib.lib()
        if gout.stride(1) != 1 or gout.stride(0) % 4 != 0 or gout.data_ptr() % 16 != 0:
            gout = gout.contiguous()
        ldg = gout.stride(0)                 # column-slice views (e.g. the dX slab) pass as is
        n = slab.size(0)
        dev = slab.device
        st = current_stream_ptr(dev)
        need_x, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        need_w = any(ctx.needs_input_grad[5:])
        mask_ptr = out.data_ptr() if out is not None else None
        ldm = out.stride(0) if out is not None else fo
        wpad, lds = slab.size(1), slab.stride(0)
        if concat:
            xs, ldxs, fi_eff, nseg = [slab], [lds], wpad, 1
        else:
            xs = [slab[:, j * fi:(j + 1) * fi] for j in range(k + 1)]
            ldxs, fi_eff, nseg = [lds] * (k + 1), fi, k + 1

        h2 = ctx.h2
        gws: List[Optional[torch.Tensor]] = [None] * (k + 1)
        gb = gx = None
        g_ptr, g_ld, g_rowmax = gout.data_ptr(), ldg, None

        dw_done = not (need_w or need_b)

        def weight_gradients():
            nonlocal gws, gb
            if need_w or need_b:
                # one output block per lins[k].weight, in either layout of the dense block
                sinks = [_grad_sink(p) for p in ctx.params] + \
                    ([_grad_sink(ctx.bias_param)] if ctx.has_bias else [])
                direct = (DIRECT_PARAM_GRAD and not torch.is_grad_enabled()
                          and all(ctx.needs_input_grad[5:]) and (need_b or not ctx.has_bias)
                          and sinks[0] is not None and all(b is sinks[0] for b in sinks))
                if direct:
                    outs = [p.grad for p in ctx.params]
                    gb_out = ctx.bias_param.grad if ctx.has_bias else None
                else:
                    outs = [torch.empty((fo, fi), dtype=torch.float32, device=dev) for _ in range(k + 1)]
                    gb_out = torch.empty(fo, dtype=torch.float32, device=dev) if need_b else None
                nbytes = L.dc_tag_linear_bwd_dw_workspace_bytes(n, fi_eff, fo, nseg)
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                args = (g_ptr, g_ld, mask_ptr, ldm, _ptr_array(xs), _i64_array(ldxs), nseg,
                        _ptr_array(outs), k + 1, fi, gb_out.data_ptr() if gb_out is not None else None,
                        int(direct), scratch.data_ptr(), nbytes, n, fi_eff, fo)
                if g_rowmax is not None and n % 16 == 0:
                    rc = L.dc_tag_linear_bwd_dw_h2(*args, g_rowmax.data_ptr(), xrowmax.data_ptr(), st)
                elif DENSE_SPLIT_BF16:
                    rc = L.dc_tag_linear_bwd_dw_split(*args, DENSE_PRODUCTS, st)
                else:
                    rc = L.dc_tag_linear_bwd_dw(*args, st)
                _lib.check(rc, "dc_tag_linear_bwd_dw")
                if direct:
                    sinks[0].note_direct_write(torch.cuda.current_stream(dev))
                else:
                    gws = [outs[j] if ctx.needs_input_grad[5 + j] else None for j in range(k + 1)]
                    gb = gb_out

        if h2 and fo % 16 == 0 and fo % 4 == 0:
            # fp16x2 path, backward in the forward's shape: gx = sum_j ((A^T)^j gm) W_j with
            # gm = g * relu' - K transposed hops on gm (which also record the row maxima), then
            # ONE dense block with the (K+1)*Fo reduction and the transposed weights.  gm and its
            # row maxima also feed dW (no mask reads there).
            gwid = (k + 1) * fo
            gslab = _alloc_slab(n, gwid, dev)
            gld = gslab.stride(0)
            g_rowmax = torch.empty(n, dtype=torch.float32, device=dev)
            hop_rowmax = torch.empty(n, dtype=torch.float32, device=dev) if need_x else None
            # (folding this pass into the transposed chain's staging was built and measured in round 4 - bit-identical, one
            # launch and 100 MB less, and 0 - 2 % SLOWER on the step: this pass runs in the shadow of the other branch's
            # dense blocks, the chain launch does not; tools/exp/hopchain_masked.hip keeps the kernel)
            _lib.check(L.dc_tag_mask_grad(gout.data_ptr(), ldg, mask_ptr, ldm, gslab.data_ptr(), gld, n,
                                          fo, g_rowmax.data_ptr(),
                                          hop_rowmax.data_ptr() if need_x else None, st),
                       "dc_tag_mask_grad")
            g_ptr, g_ld, mask_ptr = gslab.data_ptr(), gld, None
            dw_pos = _dw_position(dev)
            if need_x and dw_pos == "first":
                # dW needs gm (block 0) and the forward's slab only - not the transposed chain: on one of the two encoder
                # streams it goes in FRONT of chain + dX, so that the two branches' dW kernels do not run side by side
                # (DW_LAST_STREAMS)
                weight_gradients()
                dw_done = True
            if need_x:
                chained_hops(g, gslab, fo, k, backward=False, rowmax=hop_rowmax, transposed=True,
                             rowmax_has_block0=True)
                if dw_pos == "mid" and not dw_done:
                    weight_gradients()
                    dw_done = True
                if wt is None:                       # forward ran without needs_input_grad
                    wt = torch.empty((fi, gwid), dtype=torch.float32, device=dev)
                    wt_rowmax = torch.empty(fi, dtype=torch.float32, device=dev)
                    _lib.check(L.dc_tag_weight_prep(_ptr_array(ws), k + 1, fo, fi,
                                                    torch.empty(fo, device=dev).data_ptr(), None,
                                                    wt.data_ptr(), wt_rowmax.data_ptr(), st),
                               "dc_tag_weight_prep")
                gx = torch.empty((n, fi), dtype=torch.float32, device=dev)
                rc = L.dc_tag_linear_fwd_h2p(gslab.data_ptr(), gld, wt.data_ptr(), None, 0, gx.data_ptr(),
                                             fi, n, gwid, fi, hop_rowmax.data_ptr(), wt_rowmax.data_ptr(),
                                             None, 0, st)
                _lib.check(rc, "dc_tag_linear_fwd_h2 (dX)")
                if DEBUG_TAP is not None:
                    DEBUG_TAP(f"bwd{fi}x{fo}.hop_rowmax", hop_rowmax)
                    DEBUG_TAP(f"bwd{fi}x{fo}.g_rowmax", g_rowmax)
                    if os.environ.get("HUNT_TAP_ADJ") == "1":
                        for nm, t in (("bwd.ptr", g.bwd.ptr), ("bwd.other", g.bwd.other), ("bwd.w", g.bwd.w),
                                      ("fwd.ptr", g.fwd.ptr)):
                            DEBUG_TAP(f"bwd{fi}x{fo}.adj.{nm}", t)
                    if os.environ.get("HUNT_TAP_BIG") == "1":
                        DEBUG_TAP(f"bwd{fi}x{fo}.gx", gx)
                        DEBUG_TAP(f"bwd{fi}x{fo}.gslab", gslab)
                need_x = False                               # done

        if not dw_done:
            weight_gradients()

        if need_x:
            if concat and ctx.narrow:
                # the forward read the lins[k].weight matrices directly; the one-segment dX block wants them concatenated
                wcat = torch.empty((fo, wpad), dtype=torch.float32, device=dev)
                _lib.check(L.dc_tag_pack_weights(_ptr_array(ws), k + 1, wcat.data_ptr(), fo, fi, wpad, st),
                           "dc_tag_pack_weights")
                ws = [wcat]
            gslab = _alloc_slab(n, wpad, dev)
            gblocks = [gslab[:, j * fi:(j + 1) * fi] for j in range(k + 1)]
            gxs = [gslab] if concat else gblocks
            ldxs = [gslab.stride(0)] * len(ldxs)
            if DENSE_SPLIT_BF16:
                wsb = L.dc_tag_linear_bwd_dx_split_workspace_bytes(fi_eff, fo, nseg)
                wsx = torch.empty(wsb, dtype=torch.uint8, device=dev)
                rc = L.dc_tag_linear_bwd_dx_split(g_ptr, g_ld, mask_ptr, ldm, _ptr_array(ws), nseg,
                                                  _ptr_array(gxs), _i64_array(ldxs), wsx.data_ptr(), wsb,
                                                  n, fi_eff, fo, DENSE_PRODUCTS, st)
            else:
                rc = L.dc_tag_linear_bwd_dx(g_ptr, g_ld, mask_ptr, ldm, _ptr_array(ws), nseg,
                                            _ptr_array(gxs), _i64_array(ldxs), n, fi_eff, fo, st)
            _lib.check(rc, "dc_tag_linear_bwd_dx")
            chained_hops(g, gslab, fi, k, backward=True)  # g_{j-1} = G_{j-1} + A^T g_j
            gx = gblocks[0]
        return (None, gx, gb, None, None, *gws)


class OutInto:
    """Destination of a layer's output chosen by the caller (``tag_conv(..., next_geom=OutInto(view))``):
    a ``[N, Fo]`` row-major view, e.g. one part's rows of block 0 of a merged hop slab.  A plain object, so
    autograd does not see the buffer as an input of the layer."""

    def __init__(self, view: torch.Tensor):
        self.view = view


# --------------------------------------------------------------------------- #
# grouped TAGConv layer: both encoder branches as ONE block-diagonal launch
# --------------------------------------------------------------------------- #
_MERGED_TAG = "_dc_merged_slab"


def alloc_merged_slab(mg: GraphIndex, fi: int, k: int, dev) -> torch.Tensor:
    """``[N_total, (K+1)*fi]`` hop slab over the merged node space of ``mg`` (``GraphIndex.from_parts``)
    for a grouped TAGConv layer, tagged so that ``tag_conv_grouped`` recognises the parts' block-0 views
    (``merged_slab_part``).  The padding rows of block 0 are zeroed here (their other blocks are written
    - as zeros - by the hops: padding nodes are isolated)."""
    concat, width, wpad = tag_slab_geometry(fi, k)
    if concat:
        raise ValueError("alloc_merged_slab: wide layers only (Fi a multiple of 16, (K+1)*Fi > 128)")
    n = mg.num_nodes
    slab = _alloc_slab(n, wpad, dev, tag=("merged", n, fi, wpad, tuple(mg.row_beg), tuple(mg.rows)))
    base = slab._base if slab._base is not None else slab
    setattr(base, _SLAB_TAG, ("merged", n, fi, wpad, tuple(mg.row_beg), tuple(mg.rows)))
    ends = list(mg.row_beg[1:]) + [n]
    for r0, rows, r1 in zip(mg.row_beg, mg.rows, ends):
        if r0 + rows < r1:
            slab[r0 + rows:r1, :fi].zero_()
    return slab


def merged_slab_part(slab: torch.Tensor, mg: GraphIndex, g: int, fi: int) -> torch.Tensor:
    """Rows of part ``g`` in column block 0 of a merged slab: where that part's previous layer writes."""
    r0 = mg.row_beg[g]
    return slab[r0:r0 + mg.rows[g], :fi]


def _as_merged_slab(xs, mg: GraphIndex, fi: int, wpad: int):
    """The merged slab whose block 0 the tensors ``xs`` (one per part) are the part views of, or None."""
    base = xs[0]._base
    tag = ("merged", mg.num_nodes, fi, wpad, tuple(mg.row_beg), tuple(mg.rows))
    if base is None or getattr(base, _SLAB_TAG, None) != tag or base.dim() != 2 or not base.is_contiguous():
        return None
    ld = base.size(1)
    for g, x in enumerate(xs):
        if (x._base is not base or x.shape != (mg.rows[g], fi) or x.stride() != (ld, 1)
                or x.data_ptr() != base.data_ptr() + 4 * ld * mg.row_beg[g]):
            return None
    return base[:, :wpad] if ld > wpad else base


def grouped_eligible(fi: int, fo: int, k: int) -> bool:
    """Can ``tag_conv_grouped`` run a layer of these widths (the grouped kernels' shape limits)?"""
    return (_tag_uses_h2(fi, k) and fi == 256 and fo % 128 == 0 and fo % 16 == 0
            and ((k + 1) * fi) % 32 == 0 and ((k + 1) * fo) % 32 == 0)


def _vp_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


class _TagConvGroupedFn(torch.autograd.Function):
    """The same TAGConv layer (+ fused ReLU) of SEVERAL branches - same widths, each branch its own
    weights - over the merged node space of ``mg``: K hops over the merged adjacency (one launch each for
    all branches), ONE grouped forward block, and in backward one grouped mask kernel, K transposed merged
    hops, one grouped forward-shaped dX block, one grouped dW block + slab reduce
    (``dc_tag_grouped_*``).  Replaces the second ``conv(x, edge_index)`` call of both encoder loops of
    ``models/model.py:69-78``.  Row by row the arithmetic is that of ``_TagConvFn`` on each branch alone:
    outputs and gradients are bit-identical to it.

    ``apply(mg, relu, next_geom, ngroups, x_0..x_{G-1}, bias_0, W_0,0..W_0,K, bias_1, W_1,0.., ...)``
    returns one output per group (views of one merged buffer)."""

    @staticmethod
    def forward(ctx, mg: GraphIndex, relu: bool, next_geom, ngroups: int, *args):
        xs = args[:ngroups]
        per = (len(args) - ngroups) // ngroups
        k = per - 2
        biases = [args[ngroups + g * per] for g in range(ngroups)]
        weights = [list(args[ngroups + g * per + 1: ngroups + (g + 1) * per]) for g in range(ngroups)]
        fo, fi = weights[0][0].shape
        dev = xs[0].device
        n = mg.num_nodes
        concat, width, wpad = tag_slab_geometry(fi, k)
        if not _tag_uses_h2(fi, k) or width % 32 != 0 or any(b is None for b in biases) != all(b is None for b in biases):
            raise NotImplementedError("tag_conv_grouped: wide fp16x2 layers only, bias on all groups or none")
        L = _lib.lib()
        st = current_stream_ptr(dev)
        slab = _as_merged_slab(xs, mg, fi, wpad)
        if slab is None:
            slab = alloc_merged_slab(mg, fi, k, dev)
            for g, x in enumerate(xs):
                merged_slab_part(slab, mg, g, fi).copy_(x)
        rowmax = torch.empty(n, dtype=torch.float32, device=dev)
        chained_hops(mg, slab, fi, k, backward=False, rowmax=rowmax)
        # weights of all groups: scaled fp16x2 images (+ transposed images for dX) in one launch
        need_x = any(ctx.needs_input_grad[4:4 + ngroups])
        wmax = torch.empty((ngroups, fo), dtype=torch.float32, device=dev)
        wimg = torch.empty((ngroups, fo, width), dtype=torch.float32, device=dev)
        wt = wt_rowmax = None
        if need_x and fo % 16 == 0:
            wt = torch.empty((ngroups, fi, (k + 1) * fo), dtype=torch.float32, device=dev)
            wt_rowmax = torch.empty((ngroups, fi), dtype=torch.float32, device=dev)
        wcs = [[w.contiguous() for w in ws] for ws in weights]
        _lib.check(L.dc_tag_grouped_weight_prep(
            _vp_array([w.data_ptr() for ws in wcs for w in ws]), ngroups, k + 1, fo, fi,
            _vp_array([wmax[g].data_ptr() for g in range(ngroups)]),
            _vp_array([wimg[g].data_ptr() for g in range(ngroups)]),
            _vp_array([wt[g].data_ptr() for g in range(ngroups)]) if wt is not None else None,
            _vp_array([wt_rowmax[g].data_ptr() for g in range(ngroups)]) if wt is not None else None, st),
            "dc_tag_grouped_weight_prep")
        if isinstance(next_geom, tuple):
            nxt = alloc_merged_slab(mg, fo, next_geom[0], dev)      # next grouped layer's slab: (K_next,)
            out = nxt[:, :fo]
        else:
            out = torch.empty((n, fo), dtype=torch.float32, device=dev)
        bcs = [b.contiguous() if b is not None else None for b in biases]
        row_beg, rows = _i64_array(mg.row_beg), _i64_array(mg.rows)
        _lib.check(L.dc_tag_grouped_fwd_h2p(
            slab.data_ptr(), slab.stride(0), ngroups, row_beg, rows, n,
            _vp_array([wimg[g].data_ptr() for g in range(ngroups)]),
            _vp_array([b.data_ptr() if b is not None else None for b in bcs]), int(relu),
            out.data_ptr(), out.stride(0), width, fo, rowmax.data_ptr(),
            _vp_array([wmax[g].data_ptr() for g in range(ngroups)]), st), "dc_tag_grouped_fwd_h2p")
        ctx.mg, ctx.k, ctx.fi, ctx.fo, ctx.relu, ctx.ngroups = mg, k, fi, fo, relu, ngroups
        ctx.params, ctx.bias_params = weights, biases
        ctx.save_for_backward(slab, out if relu else None, rowmax, wt, wt_rowmax)
        outs = tuple(out[r0:r0 + r] for r0, r in zip(mg.row_beg, mg.rows))
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        slab, out, xrowmax, wt, wt_rowmax = ctx.saved_tensors
        mg, k, fi, fo, ngroups = ctx.mg, ctx.k, ctx.fi, ctx.fo, ctx.ngroups
        L = _lib.lib()
        dev = slab.device
        st = current_stream_ptr(dev)
        n = mg.num_nodes
        per = k + 2
        need_x = any(ctx.needs_input_grad[4:4 + ngroups])
        need_p = any(ctx.needs_input_grad[4 + ngroups:])
        gs = []
        for g, go in enumerate(gouts):
            if go is None:
                go = torch.zeros((mg.rows[g], fo), dtype=torch.float32, device=dev)
            if go.stride(1) != 1 or go.stride(0) % 4 != 0 or go.data_ptr() % 16 != 0:
                go = go.contiguous()
            gs.append(go)
        row_beg, rows = _i64_array(mg.row_beg), _i64_array(mg.rows)
        gwid = (k + 1) * fo
        gslab = _alloc_slab(n, gwid, dev)
        g_rowmax = torch.empty(n, dtype=torch.float32, device=dev)
        hop_rowmax = torch.empty(n, dtype=torch.float32, device=dev) if need_x else None
        _lib.check(L.dc_tag_grouped_mask_grad(
            _vp_array([t.data_ptr() for t in gs]), _i64_array([t.stride(0) for t in gs]), ngroups, row_beg, rows, n,
            out.data_ptr() if out is not None else None, out.stride(0) if out is not None else fo,
            gslab.data_ptr(), gslab.stride(0), fo, g_rowmax.data_ptr(),
            hop_rowmax.data_ptr() if need_x else None, st), "dc_tag_grouped_mask_grad")
        gxs = [None] * ngroups
        if need_x:
            if wt is None:
                raise RuntimeError("tag_conv_grouped: input gradient requested but the forward ran without it")
            chained_hops(mg, gslab, fo, k, backward=False, rowmax=hop_rowmax, transposed=True,
                         rowmax_has_block0=True)
            gx = torch.empty((n, fi), dtype=torch.float32, device=dev)
            _lib.check(L.dc_tag_grouped_fwd_h2p(
                gslab.data_ptr(), gslab.stride(0), ngroups, row_beg, rows, n,
                _vp_array([wt[g].data_ptr() for g in range(ngroups)]), None, 0, gx.data_ptr(), fi, gwid, fi,
                hop_rowmax.data_ptr(), _vp_array([wt_rowmax[g].data_ptr() for g in range(ngroups)]), st),
                "dc_tag_grouped_fwd_h2p (dX)")
            gxs = [gx[r0:r0 + r] if ctx.needs_input_grad[4 + g] else None
                   for g, (r0, r) in enumerate(zip(mg.row_beg, mg.rows))]
        pgrads = [None] * (ngroups * per)
        if need_p:
            has_bias = ctx.bias_params[0] is not None
            flat_params = [p for g in range(ngroups) for p in ([ctx.bias_params[g]] if has_bias else []) + ctx.params[g]]
            sinks = [_grad_sink(p) for p in flat_params]
            all_needed = all(ctx.needs_input_grad[4 + ngroups + g * per + j] for g in range(ngroups)
                             for j in range(per) if (j > 0 or has_bias))
            direct = (DIRECT_PARAM_GRAD and not torch.is_grad_enabled() and all_needed
                      and sinks[0] is not None and all(b is sinks[0] for b in sinks))
            if direct:
                gw_out = [[p.grad for p in ctx.params[g]] for g in range(ngroups)]
                gb_out = [ctx.bias_params[g].grad if has_bias else None for g in range(ngroups)]
            else:
                gw_out = [[torch.empty((fo, fi), dtype=torch.float32, device=dev) for _ in range(k + 1)]
                          for _ in range(ngroups)]
                gb_out = [torch.empty(fo, dtype=torch.float32, device=dev) if has_bias else None
                          for _ in range(ngroups)]
            nbytes = L.dc_tag_grouped_bwd_dw_workspace_bytes(rows, ngroups, fi, fo, k + 1)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            xblocks = [slab[:, j * fi:(j + 1) * fi] for j in range(k + 1)]
            _lib.check(L.dc_tag_grouped_bwd_dw_h2(
                gslab.data_ptr(), gslab.stride(0), _ptr_array(xblocks), _i64_array([slab.stride(0)] * (k + 1)),
                k + 1, ngroups, row_beg, rows, n,
                _vp_array([t.data_ptr() for ws in gw_out for t in ws]),
                _vp_array([t.data_ptr() if t is not None else None for t in gb_out]), int(direct),
                scratch.data_ptr(), nbytes, fi, fo, g_rowmax.data_ptr(), xrowmax.data_ptr(), st),
                "dc_tag_grouped_bwd_dw_h2")
            if direct:
                sinks[0].note_direct_write(torch.cuda.current_stream(dev))
            else:
                for g in range(ngroups):
                    base = g * per
                    if has_bias and ctx.needs_input_grad[4 + ngroups + base]:
                        pgrads[base] = gb_out[g]
                    for j in range(k + 1):
                        if ctx.needs_input_grad[4 + ngroups + base + 1 + j]:
                            pgrads[base + 1 + j] = gw_out[g][j]
        return (None, None, None, None, *gxs, *pgrads)


def tag_conv_grouped(mg: GraphIndex, xs, weights, biases, relu: bool = False, next_k=None):
    """One TAGConv layer of ``len(xs)`` branches over the merged adjacency ``mg``: ``xs[g]`` ``[rows_g, Fi]``,
    ``weights[g]`` = that branch's ``lins[0..K].weight``, ``biases[g]`` its bias.  Returns one output per
    branch.  ``next_k``: K of a grouped layer that consumes the outputs - they are then written as the part
    views of block 0 of that layer's merged slab."""
    flat = []
    xs = [resolve(x) for x in xs]
    for b, ws in zip(biases, weights):
        flat += [b] + list(ws)
    return _TagConvGroupedFn.apply(mg, bool(relu), (int(next_k),) if next_k is not None else None, len(xs),
                                   *xs, *flat)


class _SixProducts:
    """``next_geom`` marker of ``dense_linear(..., six_products=True)``."""


SIX_PRODUCTS = _SixProducts()


def dense_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 relu: bool = False, six_products: bool = False) -> torch.Tensor:
    """``act(x @ weight.T + bias)`` on the library's dense block (a TAGConv layer with K = 0: no hops):
    the ``lin`` of ``GCNConv`` / ``GATConv`` (PyG ``nn/dense/linear.py``), forward and backward.
    ``six_products``: stay on the exact three-way bf16 split (24 bits) where the three-product fp16 split (22 bits)
    would apply - GATConv's ``lin``: the gradient of its attention vectors is a sum of terms that cancel to 1 % of
    their size and sits at the parity bar already."""
    return _TagConvFn.apply(None, resolve(x), bias, bool(relu), SIX_PRODUCTS if six_products else None, weight)


def tag_conv(g: GraphIndex, x: torch.Tensor, weights, bias, relu: bool = False,
             next_geom=None) -> torch.Tensor:
    """``next_geom``: ``(width, padded width)`` of the hop slab of the TAGConv layer that consumes
    this output (None = none): the output is then allocated as column block 0 of that slab."""
    return _TagConvFn.apply(g, resolve(x), bias, bool(relu), next_geom, *weights)


# --------------------------------------------------------------------------- #
# TAGConv over bf16-stored features (BASELINE.json configs[4]): forward + backward
# --------------------------------------------------------------------------- #
_SLAB_TAG_BF16 = "_dc_hop_slab_bf16"


def _alloc_bf16(rows: int, wid: int, dev, tag=None) -> torch.Tensor:
    pad = 2 * SLAB_PAD if (SLAB_PAD > 0 and (wid * 2) % 1024 == 0) else 0      # same byte padding as fp32
    b = torch.empty((rows, wid + pad), dtype=torch.bfloat16, device=dev)
    if tag is not None:
        setattr(b, _SLAB_TAG_BF16, tag)
    return b[:, :wid] if pad else b


class _TagConvBf16Fn(torch.autograd.Function):
    """``TAGConv.forward`` (+ optional ReLU) with the features STORED as bfloat16 and fp32 master weights.
    Forward: K hops ``dc_spmm_bf16`` (bf16 rows gathered, fp32 running sum in the stable edge order, one rounding per
    stored element) into a ``[N, (K+1) * Fi]`` bf16 slab, then ONE bf16 MFMA dense block
    (``dc_tag_linear_fwd_bf16``, fp32 accumulate) with the weights rounded to bf16.  Backward, in the forward's
    shape: ``gm = g * relu'`` as bf16 (``dc_tag_mask_grad_bf16``), K TRANSPOSED bf16 hops on gm, the same dense
    block over that gradient slab with the transposed weights for ``gx``, and ``dc_tag_linear_bwd_dw_bf16`` for the
    fp32 weight / bias gradients.  What PyG reaches under ``torch.autocast(bfloat16)``, with every aggregation and
    every contraction accumulated in fp32."""

    @staticmethod
    def forward(ctx, g: GraphIndex, x: torch.Tensor, bias, relu: bool, out_dtype, next_k, *weights):
        n, fi = x.shape
        k = len(weights) - 1
        fo = weights[0].size(0)
        width = (k + 1) * fi
        dev = x.device
        L = _lib.lib()
        st = current_stream_ptr(dev)
        base = x._base
        if (base is not None and getattr(base, _SLAB_TAG_BF16, None) == (n, fi, width)
                and base.size(0) == n and base.size(1) >= width and base.is_contiguous()
                and x.data_ptr() == base.data_ptr() and x.stride() == (base.size(1), 1)):
            slab = base[:, :width] if base.size(1) > width else base   # the previous layer wrote block 0 in place
        else:
            slab = _alloc_bf16(n, width, dev)
            slab[:, :fi].copy_(x)
        for j in range(k):
            hop_bf16(g.fwd, slab[:, j * fi:(j + 1) * fi], out=slab[:, (j + 1) * fi:(j + 2) * fi],
                     weighted=g.normalize, out_dtype=torch.bfloat16)
        ws = [w.detach().contiguous() for w in weights]
        wcat = torch.empty((fo, width), dtype=torch.bfloat16, device=dev)
        _lib.check(L.dc_to_bf16(_ptr_array(ws), k + 1, fo, fi, fi, wcat.data_ptr(), width, st), "dc_to_bf16")
        if next_k is not None and out_dtype == torch.bfloat16:
            nwidth = (next_k + 1) * fo
            nxt = _alloc_bf16(n, nwidth, dev, tag=(n, fo, nwidth))
            out = nxt[:, :fo]
        else:
            out = torch.empty((n, fo), dtype=out_dtype, device=dev)
        b = bias.detach().contiguous() if bias is not None else None
        rc = L.dc_tag_linear_fwd_bf16(slab.data_ptr(), slab.stride(0), wcat.data_ptr(),
                                      b.data_ptr() if b is not None else None, int(relu), out.data_ptr(),
                                      out.stride(0), int(out.dtype == torch.bfloat16), n, width, fo, st)
        _lib.check(rc, "dc_tag_linear_fwd_bf16")
        ctx.g, ctx.k, ctx.fi, ctx.fo, ctx.relu, ctx.has_bias, ctx.x_dtype = g, k, fi, fo, relu, bias is not None, x.dtype
        ctx.save_for_backward(slab, out if relu else None, *ws)
        return out

    @staticmethod
    def backward(ctx, gout):
        slab, out, *ws = ctx.saved_tensors
        g, k, fi, fo = ctx.g, ctx.k, ctx.fi, ctx.fo
        n, dev = slab.size(0), slab.device
        L = _lib.lib()
        st = current_stream_ptr(dev)
        if gout.dtype not in (torch.bfloat16, torch.float32):
            gout = gout.float()
        if gout.stride(1) != 1:
            gout = gout.contiguous()
        need_x = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        need_w = any(ctx.needs_input_grad[6:])
        gwid = (k + 1) * fo
        gslab = _alloc_bf16(n, gwid if need_x else fo, dev)
        _lib.check(L.dc_tag_mask_grad_bf16(
            gout.data_ptr(), gout.stride(0), int(gout.dtype == torch.bfloat16),
            out.data_ptr() if out is not None else None, out.stride(0) if out is not None else fo,
            int(out is not None and out.dtype == torch.bfloat16), gslab.data_ptr(), gslab.stride(0), n, fo, st),
            "dc_tag_mask_grad_bf16")
        gx = gb = None
        gws = [None] * (k + 1)
        if need_x:
            if gwid % 32 != 0 or fo % 8 != 0:
                raise NotImplementedError("tag_conv_bf16 backward: (K+1)*Fo must be a multiple of 32")
            for j in range(k):
                hop_bf16(g.bwd, gslab[:, j * fo:(j + 1) * fo], out=gslab[:, (j + 1) * fo:(j + 2) * fo],
                         weighted=g.normalize, out_dtype=torch.bfloat16)
            wt = torch.empty((k + 1, fi, fo), dtype=torch.float32, device=dev)
            _lib.check(L.dc_tag_transpose_weights(_ptr_array(ws), k + 1, fo, fi, wt.data_ptr(), st),
                       "dc_tag_transpose_weights")
            wtcat = torch.empty((fi, gwid), dtype=torch.bfloat16, device=dev)
            _lib.check(L.dc_to_bf16(_ptr_array([wt[j] for j in range(k + 1)]), k + 1, fi, fo, fo,
                                    wtcat.data_ptr(), gwid, st), "dc_to_bf16")
            gx = torch.empty((n, fi), dtype=ctx.x_dtype, device=dev)
            _lib.check(L.dc_tag_linear_fwd_bf16(gslab.data_ptr(), gslab.stride(0), wtcat.data_ptr(), None, 0,
                                                gx.data_ptr(), fi, int(gx.dtype == torch.bfloat16), n, gwid, fi, st),
                       "dc_tag_linear_fwd_bf16 (dX)")
        if need_w or need_b:
            outs = [torch.empty((fo, fi), dtype=torch.float32, device=dev) for _ in range(k + 1)]
            gb_out = torch.empty(fo, dtype=torch.float32, device=dev) if need_b else None
            nbytes = L.dc_tag_linear_bwd_dw_bf16_workspace_bytes(n, fi, fo, k + 1)
            if nbytes < 0:
                raise NotImplementedError("tag_conv_bf16 backward: unsupported layer shape")
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.dc_tag_linear_bwd_dw_bf16(
                gslab.data_ptr(), gslab.stride(0), slab.data_ptr(), slab.stride(0), k + 1, _ptr_array(outs),
                gb_out.data_ptr() if gb_out is not None else None, 0, scratch.data_ptr(), nbytes, n, fi, fo, st),
                "dc_tag_linear_bwd_dw_bf16")
            gws = [outs[j] if ctx.needs_input_grad[6 + j] else None for j in range(k + 1)]
            gb = gb_out
        return (None, gx, gb, None, None, None, *gws)


def tag_conv_bf16(g: GraphIndex, x: torch.Tensor, weights, bias, relu: bool = False,
                  out_dtype: torch.dtype = torch.bfloat16, next_k: Optional[int] = None) -> torch.Tensor:
    """``TAGConv.forward`` (+ optional ReLU) over bfloat16-STORED features, differentiable w.r.t. ``x`` and the
    (fp32) parameters (``_TagConvBf16Fn``).  ``next_k``: K of the bf16 TAGConv layer that consumes the output - it is
    then written as column block 0 of that layer's slab (bf16 output only)."""
    _require_cuda(x, "x")
    if x.dtype != torch.bfloat16 or x.dim() != 2:
        raise ValueError("tag_conv_bf16: x must be a 2-D bfloat16 tensor")
    fi = x.size(1)
    k = len(weights) - 1
    if ((k + 1) * fi) % 32 != 0 or fi % 8 != 0:
        raise ValueError(f"tag_conv_bf16: (K+1)*Fi = {(k + 1) * fi} must be a multiple of 32 and Fi of 8")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("tag_conv_bf16: out_dtype must be bfloat16 or float32")
    grad = torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in weights)
                                        or (bias is not None and bias.requires_grad))
    fo = weights[0].size(0)
    if grad and (fo % 128 != 0 or fi % 256 != 0):
        raise NotImplementedError("tag_conv_bf16: the backward needs Fo % 128 == 0 and Fi % 256 == 0 "
                                  "(call under torch.no_grad() for other widths)")
    return _TagConvBf16Fn.apply(g, x, bias, bool(relu), out_dtype, next_k, *weights)


# --------------------------------------------------------------------------- #
# GATConv (heads = 1): edge softmax + weighted aggregation
# --------------------------------------------------------------------------- #
def _spmm_w(adj: SortedAdjacency, w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    n, f = x.shape
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dc_spmm_f32(adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr(),
                                x.data_ptr(), _rowmajor(x, "x"), None, 0, out.data_ptr(), f, n, f,
                                current_stream_ptr(x.device))
    _lib.check(rc, "dc_spmm_f32")
    return out


class _GatAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g: GraphIndex, h, a_src, a_dst, slope: float):
        L = _lib.lib()
        h, a_src, a_dst = h.contiguous(), a_src.contiguous(), a_dst.contiguous()
        n = h.size(0)
        st = current_stream_ptr(h.device)
        alpha = torch.zeros(max(g.capacity, 1), dtype=torch.float32, device=h.device)
        _lib.check(L.dc_gat_edge_softmax_fwd(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(),
                                             a_src.data_ptr(), a_dst.data_ptr(), slope,
                                             alpha.data_ptr(), n, st), "dc_gat_edge_softmax_fwd")
        out = _spmm_w(g.fwd, alpha, h)
        ctx.g, ctx.slope = g, slope
        ctx.save_for_backward(h, a_src, a_dst, alpha)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        h, a_src, a_dst, alpha = ctx.saved_tensors
        g, slope = ctx.g, ctx.slope
        gout = gout.contiguous()
        n, f = h.shape
        dev = h.device
        st = current_stream_ptr(dev)
        cap = max(g.capacity, 1)
        b2f = g.bwd_to_fwd()
        cnt = g.fwd.ptr[-1:]
        # d out / d h : transposed aggregation with alpha re-ordered by source
        alpha_b = torch.zeros(cap, dtype=torch.float32, device=dev)
        _lib.check(L.dc_gather_f32(alpha.data_ptr(), b2f.data_ptr(), alpha_b.data_ptr(),
                                   cnt.data_ptr(), g.capacity, st), "dc_gather_f32")
        gh = _spmm_w(g.bwd, alpha_b, gout)
        # d out / d alpha
        galpha = torch.zeros(cap, dtype=torch.float32, device=dev)
        _lib.check(L.dc_sddmm_f32(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), gout.data_ptr(), f,
                                  h.data_ptr(), f, galpha.data_ptr(), n, f, st), "dc_sddmm_f32")
        ge = torch.zeros(cap, dtype=torch.float32, device=dev)
        g_a_dst = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.dc_gat_edge_softmax_bwd(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(),
                                             a_src.data_ptr(), a_dst.data_ptr(), slope,
                                             alpha.data_ptr(), galpha.data_ptr(), ge.data_ptr(),
                                             g_a_dst.data_ptr(), n, st), "dc_gat_edge_softmax_bwd")
        g_a_src = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.dc_segment_sum_f32(g.bwd.ptr.data_ptr(), b2f.data_ptr(), ge.data_ptr(),
                                        g_a_src.data_ptr(), n, st), "dc_segment_sum_f32")
        return None, gh, g_a_src, g_a_dst, None


def gat_aggregate(g: GraphIndex, h, a_src, a_dst, slope: float) -> torch.Tensor:
    return _GatAggregateFn.apply(g, h, a_src, a_dst, float(slope))



# --------------------------------------------------------------------------- #
# GCNConv / GATConv: aggregation + bias + ReLU as one node, row-wise passes fused (dc_gnn_epi.hip)
# --------------------------------------------------------------------------- #
FUSED_GNN_EPILOGUE = True          # (False: the unfused layer composition - tests compare the two)


def fused_gnn_ok(h: torch.Tensor) -> bool:
    """Widths / layouts the fused GCN / GAT layer kernels take (F % 4 == 0, F / 4 divides 256, aligned rows)."""
    f = h.size(1)
    return (FUSED_GNN_EPILOGUE and h.is_cuda and h.dtype == torch.float32 and f % 4 == 0 and 4 <= f <= 1024
            and 256 % (f // 4) == 0)


def _agg_bias_act(adj: SortedAdjacency, w: torch.Tensor, h: torch.Tensor, bias, relu: bool) -> torch.Tensor:
    n, f = h.shape
    y = torch.empty((n, f), dtype=torch.float32, device=h.device)
    _lib.check(_lib.lib().dc_spmm_f32_bias_act(adj.ptr.data_ptr(), adj.other.data_ptr(), w.data_ptr(), h.data_ptr(),
                                               _rowmajor(h, "h"), bias.data_ptr() if bias is not None else None,
                                               int(relu), y.data_ptr(), f, n, f, current_stream_ptr(h.device)),
               "dc_spmm_f32_bias_act")
    return y


def _mask_and_bias_grad(gy: torch.Tensor, y: Optional[torch.Tensor], bias_param, need_bias: bool):
    """gm = gy * (y > 0) (y None: gm is gy) and the bias gradient sum_i gm[i, :] in one pass.  The bias gradient
    goes straight into the bucket's view in direct mode (-> None), else it is returned."""
    n, f = gy.shape
    dev = gy.device
    if y is None and not need_bias:
        return gy, None
    L = _lib.lib()
    gm = torch.empty((n, f), dtype=torch.float32, device=dev) if y is not None else None
    sink = _grad_sink(bias_param) if (need_bias and bias_param is not None) else None
    direct = DIRECT_PARAM_GRAD and not torch.is_grad_enabled() and sink is not None
    if need_bias:
        gb = bias_param.grad if direct else torch.empty(f, dtype=torch.float32, device=dev)
    else:
        gb = torch.empty(f, dtype=torch.float32, device=dev)         # (the kernel always forms it: one pass either way)
    nb = L.dc_colsum_workspace_bytes(n, f, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    _lib.check(L.dc_mask_colsum_f32(gy.data_ptr(), gy.stride(0), y.data_ptr() if y is not None else None,
                                    y.stride(0) if y is not None else 0, gm.data_ptr() if gm is not None else None,
                                    gm.stride(0) if gm is not None else 0, n, f, ws.data_ptr(), ws.numel(),
                                    gb.data_ptr(), int(direct), current_stream_ptr(dev)), "dc_mask_colsum_f32")
    if direct:
        sink.note_direct_write(torch.cuda.current_stream(dev))
    return (gm if gm is not None else gy), (None if (direct or not need_bias) else gb)


class _GcnAggFn(torch.autograd.Function):
    """``act(A_hat h + bias)`` of a GCNConv layer (PyG gcn_conv.py: ``propagate`` + ``out + bias``, then the encoder's
    ReLU, models/model.py:71,77) as ONE aggregation launch; backward: mask + bias gradient in one pass, then the
    transposed aggregation."""

    @staticmethod
    def forward(ctx, g: GraphIndex, h, bias, relu: bool):
        h = h.contiguous()
        y = _agg_bias_act(g.fwd, g.fwd.w, h, bias, relu)
        ctx.g, ctx.relu, ctx.bias_param = g, relu, bias
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        need_b = ctx.bias_param is not None and ctx.needs_input_grad[2]
        gm, gb = _mask_and_bias_grad(gy, y, ctx.bias_param, need_b)
        gh = hop(ctx.g.bwd, gm, weighted=True) if ctx.needs_input_grad[1] else None
        return None, gh, gb, None


def gcn_aggregate(g: GraphIndex, h: torch.Tensor, bias, relu: bool = False) -> torch.Tensor:
    return _GcnAggFn.apply(g, resolve(h), bias, bool(relu))


class _GatConvFn(torch.autograd.Function):
    """Everything of a GATConv layer (heads = 1) behind its ``lin``: both attention dot products in one pass over h,
    edge softmax, ``act(sum_j a_ij h_j + bias)`` as one aggregation launch; backward: mask + bias gradient in one
    pass, transposed aggregation, SDDMM + softmax backward, and the dot products' backward (rank-one updates of dh
    and the two attention-vector gradients) in one pass (PyG gat_conv.py, utils/_softmax.py)."""

    @staticmethod
    def forward(ctx, g: GraphIndex, h, att_src, att_dst, bias, slope: float, relu: bool):
        L = _lib.lib()
        h = h.contiguous()
        n, f = h.shape
        dev = h.device
        st = current_stream_ptr(dev)
        a_s, a_d = att_src.reshape(-1).contiguous(), att_dst.reshape(-1).contiguous()
        a_src = torch.empty(n, dtype=torch.float32, device=dev)
        a_dst = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.dc_gat_alpha_fwd(h.data_ptr(), f, a_s.data_ptr(), a_d.data_ptr(), a_src.data_ptr(),
                                      a_dst.data_ptr(), n, f, st), "dc_gat_alpha_fwd")
        alpha = torch.zeros(max(g.capacity, 1), dtype=torch.float32, device=dev)
        _lib.check(L.dc_gat_edge_softmax_fwd(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), a_src.data_ptr(),
                                             a_dst.data_ptr(), slope, alpha.data_ptr(), n, st),
                   "dc_gat_edge_softmax_fwd")
        y = _agg_bias_act(g.fwd, alpha, h, bias, relu)
        ctx.g, ctx.slope, ctx.relu = g, slope, relu
        ctx.params = (att_src, att_dst, bias)
        ctx.save_for_backward(h, a_src, a_dst, alpha, a_s, a_d, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        h, a_src, a_dst, alpha, a_s, a_d, y = ctx.saved_tensors
        g, slope = ctx.g, ctx.slope
        att_src, att_dst, bias = ctx.params
        gy = gy.contiguous()
        n, f = h.shape
        dev = h.device
        st = current_stream_ptr(dev)
        need_b = bias is not None and ctx.needs_input_grad[4]
        gm, gb = _mask_and_bias_grad(gy, y, bias, need_b)
        cap = max(g.capacity, 1)
        b2f = g.bwd_to_fwd()
        cnt = g.fwd.ptr[-1:]
        alpha_b = torch.zeros(cap, dtype=torch.float32, device=dev)
        _lib.check(L.dc_gather_f32(alpha.data_ptr(), b2f.data_ptr(), alpha_b.data_ptr(), cnt.data_ptr(), g.capacity, st),
                   "dc_gather_f32")
        gh = _spmm_w(g.bwd, alpha_b, gm)
        galpha = torch.zeros(cap, dtype=torch.float32, device=dev)
        _lib.check(L.dc_sddmm_f32(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), gm.data_ptr(), f, h.data_ptr(), f,
                                  galpha.data_ptr(), n, f, st), "dc_sddmm_f32")
        ge = torch.zeros(cap, dtype=torch.float32, device=dev)
        g_a_dst = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.dc_gat_edge_softmax_bwd(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), a_src.data_ptr(),
                                             a_dst.data_ptr(), slope, alpha.data_ptr(), galpha.data_ptr(),
                                             ge.data_ptr(), g_a_dst.data_ptr(), n, st), "dc_gat_edge_softmax_bwd")
        g_a_src = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.dc_segment_sum_f32(g.bwd.ptr.data_ptr(), b2f.data_ptr(), ge.data_ptr(), g_a_src.data_ptr(), n, st),
                   "dc_segment_sum_f32")
        # the attention dot products' backward: gh += ga_src att_src + ga_dst att_dst, the two vector gradients
        sinks = [_grad_sink(att_src), _grad_sink(att_dst)]
        direct = (DIRECT_PARAM_GRAD and not torch.is_grad_enabled() and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
                  and sinks[0] is not None and sinks[1] is sinks[0])
        if direct:
            gs, gd = att_src.grad.view(-1), att_dst.grad.view(-1)
        else:
            gs = torch.empty(f, dtype=torch.float32, device=dev)
            gd = torch.empty(f, dtype=torch.float32, device=dev)
        nb = L.dc_colsum_workspace_bytes(n, f, 2)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        _lib.check(L.dc_gat_alpha_bwd(h.data_ptr(), f, g_a_src.data_ptr(), g_a_dst.data_ptr(), a_s.data_ptr(),
                                      a_d.data_ptr(), gh.data_ptr(), f, n, f, ws.data_ptr(), ws.numel(), gs.data_ptr(),
                                      gd.data_ptr(), int(direct), st), "dc_gat_alpha_bwd")
        if direct:
            sinks[0].note_direct_write(torch.cuda.current_stream(dev))
            gs = gd = None
        else:
            gs, gd = gs.view_as(att_src), gd.view_as(att_dst)
        return None, gh, gs, gd, gb, None, None


def gat_conv(g: GraphIndex, h, att_src, att_dst, bias, slope: float, relu: bool = False) -> torch.Tensor:
    return _GatConvFn.apply(g, resolve(h), att_src, att_dst, bias, float(slope), bool(relu))

# --------------------------------------------------------------------------- #
# the two training losses in one pass (train.py:51-53, models/losses.py:7-19)
# --------------------------------------------------------------------------- #
class _ContactLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g: GraphIndex, pred: torch.Tensor, target: torch.Tensor):
        L = _lib.lib()
        n = pred.size(0)
        dev = pred.device
        pred_c = pred if pred.stride(1) == 1 else pred.contiguous()
        tgt_c = target if target.stride(1) == 1 else target.contiguous()
        g1 = torch.empty((n, 3), dtype=torch.float32, device=dev)
        g2 = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        nb = L.dc_contact_loss_workspace_bytes(n)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        rc = L.dc_contact_loss(g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.bwd.ptr.data_ptr(),
                               g.bwd.other.data_ptr(), pred_c.data_ptr(), pred_c.stride(0),
                               tgt_c.data_ptr(), tgt_c.stride(0), n, g.num_input_edges, g1.data_ptr(),
                               g2.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, current_stream_ptr(dev))
        _lib.check(rc, "dc_contact_loss")
        ctx.save_for_backward(g1, g2)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_gcl):
        g1, g2 = ctx.saved_tensors
        return None, g1 * g_l1 + g2 * g_gcl, None


def contact_losses(g: GraphIndex, pred_pos: torch.Tensor, target_pos: torch.Tensor):
    """``(L1Loss(pred, target), GradientConsistencyLoss(pred, target))`` over the edge set of ``g``
    (``train.py:51-53``) in one node pass, differentiable w.r.t. ``pred_pos`` (the target is data)."""
    pred_pos, target_pos = resolve(pred_pos), resolve(target_pos)
    _require_cuda(pred_pos, "pred_pos")
    for name, t in (("pred_pos", pred_pos), ("target_pos", target_pos)):
        if t.dim() != 2 or t.size(1) != 3 or t.dtype != torch.float32:
            raise ValueError(f"contact_losses: {name} must be float32 [N, 3]")
    if pred_pos.shape != target_pos.shape or pred_pos.size(0) != g.num_nodes:
        raise ValueError("contact_losses: pred / target / graph sizes differ")
    if g.self_loops:
        raise ValueError("contact_losses: needs the adjacency of the raw edge set (no self-loop rewriting)")
    if pred_pos.size(0) == 0:
        raise ValueError("contact_losses: empty graph")
    return _ContactLossFn.apply(g, pred_pos, target_pos)
