"""Sorted adjacency of a (batched) ``edge_index`` on the GPU, built once, reused.

PyG's ``TAGConv``/``GCNConv`` recompute ``gcn_norm`` and re-scatter over the raw
``edge_index`` in every conv call (``/root/reference/models/model.py:71,77``: 4 conv
calls, 12 hops per forward).  Here the topology work is done once per
``edge_index`` tensor and shared by every hop of every layer, forward and
backward:

* ``fwd``  - edges grouped by destination, stable (``np.argsort(dst, 'stable')``):
  ``ptr [N+1]``, ``other`` = source ids, ``perm`` = original edge ids, ``w`` =
  ``gcn_norm`` weights in that order;
* ``bwd``  - the same edge set grouped by source (the transposed operator used by
  the backward hop), with the same weights re-ordered.

The cache is keyed on the tensor's storage address + version + shape, and keeps
the ``edge_index`` tensor alive so the address cannot be recycled.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"deformcontact_amd: {what} must live on a HIP device (got {t.device}). "
            "There is no CPU path in this package; the CPU oracle under oracle/ is test-only.")


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


@dataclass
class SortedAdjacency:
    ptr: torch.Tensor               # int32 [N+1]
    other: torch.Tensor             # int32 [cap]
    perm: torch.Tensor              # int32 [cap]
    w: Optional[torch.Tensor]       # fp32  [cap] or None


class GraphIndex:
    """CSR (by destination) + transposed (by source) views of one edge set."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, *, self_loops: bool = False,
                 normalize: bool = True, validate: bool = False):
        _require_cuda(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        self.edge_index = edge_index.contiguous()
        self.num_nodes = int(num_nodes)
        self.num_input_edges = int(edge_index.size(1))
        self.self_loops = bool(self_loops)
        self.normalize = bool(normalize)
        self.device = edge_index.device
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fwd = self._build(key_row=1, deg_ptr=None)
        self.bwd = self._build(key_row=0, deg_ptr=self.fwd.ptr)
        self._pos_fwd = None
        self._segments = False          # not computed yet
        self._bwd_to_fwd = None
        self._num_edges = None if self_loops else self.num_input_edges
        if validate:
            self.validate()

    # capacity of the per-edge arrays (upper bound on E' when loops are appended)
    @property
    def capacity(self) -> int:
        return self.num_input_edges + (self.num_nodes if self.self_loops else 0)

    @property
    def num_edges(self) -> int:
        """Actual number of edges after self-loop rewriting (syncs once)."""
        if self._num_edges is None:
            self._num_edges = int(self.fwd.ptr[-1].item())
        return self._num_edges

    def validate(self) -> None:
        """Raise if any endpoint was outside ``[0, num_nodes)`` (synchronises)."""
        if int(self._status.item()) != 0:
            raise IndexError(
                f"edge_index contains node ids outside [0, {self.num_nodes}) "
                "(PyG/ATen would raise an index error here)")

    def _build(self, key_row: int, deg_ptr) -> SortedAdjacency:
        L = _lib.lib()
        dev, n, e = self.device, self.num_nodes, self.num_input_edges
        cap = max(self.capacity, 1)
        ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        other = torch.empty(cap, dtype=torch.int32, device=dev)
        perm = torch.empty(cap, dtype=torch.int32, device=dev)
        w = torch.empty(cap, dtype=torch.float32, device=dev) if self.normalize else None
        ws_bytes = L.dc_csr_workspace_bytes(e, n)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        rc = L.dc_csr_build(self.edge_index.data_ptr(), e, n, key_row, int(self.self_loops),
                            ptr.data_ptr(), other.data_ptr(), perm.data_ptr(),
                            deg_ptr.data_ptr() if deg_ptr is not None else None,
                            w.data_ptr() if w is not None else None,
                            self._status.data_ptr(), ws.data_ptr(), ws.numel(),
                            current_stream_ptr(dev))
        _lib.check(rc, "dc_csr_build")
        return SortedAdjacency(ptr, other, perm, w)

    def segments(self):
        """Node ranges no edge leaves (one or more whole meshes of the batch each), merged greedily
        up to the LDS capacity of ``dc_multihop_f32``; ``None`` when some connected block is too
        large (then the hops run one launch at a time).  Computed once (synchronises)."""
        if self._segments is not False:
            return self._segments
        L = _lib.lib()
        cap_n, cap_e = L.dc_multihop_max_segment_nodes(), L.dc_multihop_max_segment_edges()
        n, ei = self.num_nodes, self.edge_index
        self._segments = None
        if n == 0:
            return None
        idx = torch.arange(n, device=self.device)
        hi, lo = idx.clone(), idx.clone()
        if ei.size(1) > 0:
            a, b = ei[0].clamp(0, n - 1), ei[1].clamp(0, n - 1)
            hi.scatter_reduce_(0, a, b, "amax").scatter_reduce_(0, b, a, "amax")
            lo.scatter_reduce_(0, a, b, "amin").scatter_reduce_(0, b, a, "amin")
        pm = torch.cummax(hi, 0).values                      # max reach of nodes 0..i
        sm = torch.cummin(lo.flip(0), 0).values.flip(0)      # min reach of nodes i..n-1
        ok = (pm[:-1] < idx[1:]) & (sm[1:] >= idx[1:])       # a cut before node b is clean
        cuts = [0] + (torch.nonzero(ok).flatten() + 1).tolist() + [n]
        eptr = self.fwd.ptr[torch.tensor(cuts, device=self.device)].tolist()
        seg, start, nodes, edges = [0], 0, 0, 0
        for i in range(len(cuts) - 1):
            cn, ce = cuts[i + 1] - cuts[i], eptr[i + 1] - eptr[i]
            if cn > cap_n or ce > cap_e:
                return None
            if nodes + cn > cap_n or edges + ce > cap_e:
                seg.append(cuts[i])
                nodes, edges = 0, 0
            nodes, edges = nodes + cn, edges + ce
        seg.append(n)
        self._segments = (torch.tensor(seg, dtype=torch.int32, device=self.device), len(seg) - 1)
        return self._segments

    def pos_in_fwd(self) -> torch.Tensor:
        """``pos[e]`` = position of edge id ``e`` in the destination-sorted order."""
        if self._pos_fwd is None:
            L = _lib.lib()
            cap = max(self.capacity, 1)
            pos = torch.full((cap,), -1, dtype=torch.int32, device=self.device)
            rc = L.dc_invert_perm(self.fwd.perm.data_ptr(), self.fwd.ptr[-1:].data_ptr(),
                                  pos.data_ptr(), self.capacity, current_stream_ptr(self.device))
            _lib.check(rc, "dc_invert_perm")
            self._pos_fwd = pos
        return self._pos_fwd

    def bwd_to_fwd(self) -> torch.Tensor:
        """For every source-sorted edge, its position in the destination-sorted order."""
        if self._bwd_to_fwd is None:
            L = _lib.lib()
            cap = max(self.capacity, 1)
            out = torch.zeros(cap, dtype=torch.int32, device=self.device)
            rc = L.dc_compose_perm(self.pos_in_fwd().data_ptr(), self.bwd.perm.data_ptr(),
                                   out.data_ptr(), self.fwd.ptr[-1:].data_ptr(), self.capacity,
                                   current_stream_ptr(self.device))
            _lib.check(rc, "dc_compose_perm")
            self._bwd_to_fwd = out
        return self._bwd_to_fwd


# --------------------------------------------------------------------------- #
# cache: one GraphIndex per (edge_index storage, version, shape, N, flags)
# --------------------------------------------------------------------------- #
_CACHE: "OrderedDict[tuple, GraphIndex]" = OrderedDict()
_CACHE_MAX = 32


def graph_index(edge_index: torch.Tensor, num_nodes: int, *, self_loops: bool = False,
                normalize: bool = True) -> GraphIndex:
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
           tuple(edge_index.stride()), int(num_nodes), bool(self_loops), bool(normalize),
           edge_index.device.index)
    g = _CACHE.get(key)
    if g is not None:
        _CACHE.move_to_end(key)
        return g
    g = GraphIndex(edge_index, num_nodes, self_loops=self_loops, normalize=normalize)
    g._src_ref = edge_index          # keeps the storage (and its address) alive
    _CACHE[key] = g
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return g


def clear_cache() -> None:
    _CACHE.clear()
