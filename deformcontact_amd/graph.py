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

import os

from . import _lib

#: ``DC_VALIDATE=1``: every new ``GraphIndex`` checks its node ids on the host (one device sync per
#: new ``edge_index``) and raises ``IndexError`` as PyG / ATen would; off by default because the
#: check would serialise the training stream (out-of-range edges are skipped and flagged either way).
VALIDATE = os.environ.get("DC_VALIDATE", "0") == "1"


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"deformcontact_amd: {what} must live on a HIP device (got {t.device}). "
            "There is no CPU path in this package; the CPU oracle under oracle/ is test-only.")


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def capture_id(device) -> int:
    """Sequence id of the hipGraph capture the current stream takes part in (0 = not capturing)."""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    return int(_lib.lib().dc_stream_capture_id(current_stream_ptr(device)))


@dataclass
class SortedAdjacency:
    ptr: torch.Tensor               # int32 [N+1]
    other: torch.Tensor             # int32 [cap]
    perm: torch.Tensor              # int32 [cap]
    w: Optional[torch.Tensor]       # fp32  [cap] or None
    #: > 0: a ROW WINDOW of a merged adjacency (``GraphIndex.from_parts``): ``ptr`` is a slice of the merged
    #: ``ptr`` and ``other`` holds ids of the merged node space - the feature matrices the hop sees hold only
    #: the window's rows, neighbour row = ``other[p] - row_offset`` (``dc_spmm_f32_window``)
    row_offset: int = 0


#: every group of a merged node space starts at a multiple of this many rows (``DC_GROUP_ALIGN`` in
#: include/deformcontact.h: no 128 / 256-row tile and no dW node chunk straddles two groups)
GROUP_ALIGN = 256


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class GraphIndex:
    """CSR (by destination) + transposed (by source) views of one edge set."""

    def __init__(self, edge_index: Optional[torch.Tensor], num_nodes: int, *, self_loops: bool = False,
                 normalize: bool = True, validate: bool = False, parts=None, segments=None):
        #: ``(node offsets, edge offsets, B)`` host arrays when the one-launch segmented build applies
        self._segments = None
        #: merged (block-diagonal) adjacency: ``[(edge_index, num_nodes), ...]``; see ``from_parts``
        self.parts = None
        if parts is not None:
            if not 1 <= len(parts) <= 4:
                raise ValueError("GraphIndex.from_parts: 1..4 parts (DC_MAX_PARTS)")
            eis = []
            for ei, n in parts:
                _require_cuda(ei, "edge_index")
                if ei.dtype != torch.int64 or ei.dim() != 2 or ei.size(0) != 2:
                    raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
                eis.append(ei.contiguous())
            self.parts = [(ei, int(n)) for ei, (_, n) in zip(eis, parts)]
            self.rows = [n for _, n in self.parts]
            self.row_beg, beg = [], 0
            for n in self.rows:
                self.row_beg.append(beg)
                beg += _round_up(n, GROUP_ALIGN)
            self.edge_beg = [0]
            for ei, _ in self.parts:
                self.edge_beg.append(self.edge_beg[-1] + int(ei.size(1)))
            num_nodes = beg                                  # padded total (multiple of GROUP_ALIGN)
            edge_index = None
            self.edge_index = None
            self.num_input_edges = self.edge_beg[-1]
            self.device = eis[0].device
        else:
            _require_cuda(edge_index, "edge_index")
            if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
                raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
            self.edge_index = edge_index.contiguous()
            self.num_input_edges = int(edge_index.size(1))
            self.device = edge_index.device
        self.num_nodes = int(num_nodes)
        self.self_loops = bool(self_loops)
        self.normalize = bool(normalize)
        #: set by callers that guarantee the topology is constant for the life of a captured graph
        #: (bench.py's cached-topology mode): lets ``graph_index`` reuse this entry under capture
        self._static_ok = False
        self._capture_id = capture_id(self.device)
        # the build's status word (a flag OR-ed in when an endpoint is out of range).  A private, cleared word costs a fill
        # launch at the head of every new batch's build (5 us on each branch's critical path of the captured step); when
        # nobody has asked for validation the one-launch segmented build (which only ORs) takes a word SHARED by all such
        # graphs of the device, and `validate()` - should it be called after all - re-runs the build with a private one
        self._status_shared = False
        self._status = None
        if (segments is not None and parts is None and not self_loops and not validate and not VALIDATE):
            self._status = _shared_status(self.device)
            self._status_shared = self._status is not None
        if self._status is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        dev, n, cap = self.device, self.num_nodes, max(self.capacity, 1)

        def side():
            return SortedAdjacency(
                torch.empty(n + 1, dtype=torch.int32, device=dev),
                torch.empty(cap, dtype=torch.int32, device=dev),
                torch.empty(cap, dtype=torch.int32, device=dev),
                torch.empty(cap, dtype=torch.float32, device=dev) if self.normalize else None)

        self.fwd, self.bwd = side(), side()
        nbytes = _lib.lib().dc_graph_workspace_bytes(self.num_input_edges, n)
        self._workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        self._pos_fwd = None
        self._bwd_to_fwd = None
        self._num_edges = None if self_loops else self.num_input_edges
        #: nodes of the largest graph of the batch (``ops.chained_hops`` picks ``dc_hop_chain_f32`` by it)
        self._seg_max_nodes = 0
        #: ``(node offsets, B)`` of the batch's graphs - all ``dc_hop_chain_f32`` needs of the layout (it holds for
        #: graphs beyond the segmented BUILD's LDS caps too: a 4,096-node mesh has ~24k edges, the build takes 16k)
        self._layout = None
        if segments is not None and parts is None and not self_loops:
            self._segments = _segment_arrays(segments, self.num_nodes, self.num_input_edges, self.device)
            self._layout = _layout_arrays(segments, self.num_nodes)
        elif parts is not None and not self_loops:
            # merged node space: when every part's edge_index carries its batch layout (data.Batch._tag_edge_layout), the
            # merged space is a union of those graphs plus the isolated padding rows between the parts, which enter as graphs
            # without edges - dc_hop_chain_f32 then runs the merged hops as one launch too (round 6; until then the merged
            # encoder path hopped one launch per hop: 6 x 35 us against 2 chain launches)
            nodes = [0]
            for g, (ei, n) in enumerate(self.parts):
                lay = edge_layout(ei)
                if lay is None or int(lay[0][-1]) != n or int(lay[1][-1]) != int(ei.size(1)):
                    nodes = None
                    break
                nodes += [self.row_beg[g] + int(v) for v in lay[0][1:]]
                end = self.row_beg[g + 1] if g + 1 < len(self.parts) else self.num_nodes
                if nodes[-1] < end:
                    nodes.append(end)                       # the part's padding rows: a graph of isolated nodes
            if nodes is not None:
                self._layout = _layout_arrays((nodes,), self.num_nodes)
        if self._layout is not None:
            nodes = self._layout[0]
            self._seg_max_nodes = max(nodes[i + 1] - nodes[i] for i in range(self._layout[1]))
        self.rebuild()
        if validate:
            self.validate()
        elif VALIDATE:
            if self._capture_id != 0:
                # built under hipGraph capture: the build has not RUN and a host read-back would invalidate
                # the capture - the status word is checked by the first validate_pending() after a replay
                _PENDING_VALIDATION.append(self)
            else:
                self.validate()

    def rebuild(self) -> None:
        """(Re)run the build pipeline (``dc_graph_build``: both sides, 5-7 launches, no host sync)
        on the current stream into this object's buffers - e.g. after the ``edge_index`` buffer
        has been refilled with a new batch of the same shape.  Legal under hipGraph capture."""
        f, t = self.fwd, self.bwd
        out = (f.ptr.data_ptr(), f.other.data_ptr(), f.perm.data_ptr(),
               f.w.data_ptr() if f.w is not None else None,
               t.ptr.data_ptr(), t.other.data_ptr(), t.perm.data_ptr(),
               t.w.data_ptr() if t.w is not None else None,
               self._status.data_ptr(), self._workspace.data_ptr(), self._workspace.numel(),
               current_stream_ptr(self.device))
        if self._segments is not None:
            nptr, eptr, nseg = self._segments
            rc = _lib.lib().dc_graph_build_segmented(
                self.edge_index.data_ptr(), self.num_input_edges, self.num_nodes, nptr, eptr, nseg,
                *out[:9], out[11])
            _lib.check(rc, "dc_graph_build_segmented")
        elif self.parts is not None:
            import ctypes
            k = len(self.parts)
            rc = _lib.lib().dc_graph_build_parts(
                (ctypes.c_void_p * k)(*[ei.data_ptr() for ei, _ in self.parts]),
                (ctypes.c_int64 * k)(*[int(ei.size(1)) for ei, _ in self.parts]),
                (ctypes.c_int64 * k)(*self.row_beg), (ctypes.c_int64 * k)(*self.rows), k,
                self.num_nodes, int(self.self_loops), *out)
            _lib.check(rc, "dc_graph_build_parts")
        else:
            rc = _lib.lib().dc_graph_build(
                self.edge_index.data_ptr(), self.num_input_edges, self.num_nodes, int(self.self_loops), *out)
            _lib.check(rc, "dc_graph_build")
        self._pos_fwd = self._bwd_to_fwd = None
        if self.self_loops:
            self._num_edges = None

    @classmethod
    def from_parts(cls, parts, *, self_loops: bool = False, normalize: bool = True) -> "GraphIndex":
        """ONE sorted adjacency over the block-diagonal union of several graphs - ``parts`` =
        ``[(edge_index, num_nodes), ...]`` (e.g. the soft and the rigid graph of a batch,
        ``train.py:36-38``) - built by ``dc_graph_build_parts`` without materialising the merged
        ``edge_index``.  Part ``g`` owns rows ``[row_beg[g], row_beg[g] + rows[g])`` of the merged node
        space; every part starts at a multiple of ``GROUP_ALIGN`` rows, the rows in between are
        isolated padding nodes and ``num_nodes`` is the padded total.  Rows of a part come out exactly
        as its own ``GraphIndex`` would hold them (same stable order, same ``gcn_norm`` weights), so a
        hop over the merged adjacency is bit-identical, row by row, to the per-part hops - in one
        launch.  ``window(g)`` is the per-part view for code that still works part by part."""
        return cls(None, 0, self_loops=self_loops, normalize=normalize, parts=parts)

    def window(self, g: int) -> "GraphWindow":
        """Part ``g`` of a merged adjacency as a graph of its own (shares the merged arrays)."""
        if self.parts is None:
            raise ValueError("window(): not a merged adjacency")
        return GraphWindow(self, g)

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
        if self._status_shared:
            # built with the device's shared status word (nobody had asked for validation): build again, same buffers,
            # same result, with a word of its own
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._status_shared = False
            self.rebuild()
        if int(self._status.item()) != 0:
            self._status.zero_()
            raise IndexError(
                f"edge_index contains node ids outside [0, {self.num_nodes}) "
                "(PyG/ATen would raise an index error here)")

    def tensors(self):
        """Every device buffer this object owns (for ``record_stream`` when it is built on one
        stream and consumed on another)."""
        out = [self._status, self._workspace]
        for adj in (self.fwd, self.bwd):
            out += [t for t in (adj.ptr, adj.other, adj.perm, adj.w) if t is not None]
        for entry in getattr(self, "_hop_cache", {}).values():
            out += [t for t in entry[:2] if t is not None]
        return out

    def record_stream(self, stream) -> None:
        for t in self.tensors():
            t.record_stream(stream)

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


#: per-graph caps of the one-launch segmented build (include/deformcontact.h DC_SEG_MAX_NODES / DC_SEG_MAX_EDGES)
SEG_MAX_NODES, SEG_MAX_EDGES = 4096, 16384
SEGMENTED_BUILD = True
def _segment_arrays(segments, num_nodes: int, num_edges: int, device):
    """Validate the layout of a batch - ``segments = (node offsets, edge offsets)``, two ascending host sequences
    of B + 1 ints as ``Batch.from_data_list`` records them - and return it as the host arrays
    ``dc_graph_build_segmented`` takes, or None when the layout does not qualify (a graph beyond the LDS caps,
    offsets that do not cover [0, N] / [0, E], an empty edge set): the caller then runs the global pipeline."""
    if not SEGMENTED_BUILD or num_edges <= 0 or num_nodes <= 0:
        return None
    nodes, edges = tuple(int(v) for v in segments[0]), tuple(int(v) for v in segments[1])
    nseg = len(nodes) - 1
    if nseg < 1 or len(edges) != nseg + 1 or nodes[0] != 0 or edges[0] != 0 or nodes[-1] != num_nodes \
            or edges[-1] != num_edges:
        return None
    for i in range(nseg):
        dn, de = nodes[i + 1] - nodes[i], edges[i + 1] - edges[i]
        if not (0 <= dn <= SEG_MAX_NODES and 0 <= de <= SEG_MAX_EDGES) or (dn == 0 and de > 0):
            return None                      # (edges without nodes: the global pipeline flags them)
    import ctypes
    return ((ctypes.c_int64 * (nseg + 1))(*nodes), (ctypes.c_int64 * (nseg + 1))(*edges), nseg)


def _layout_arrays(segments, num_nodes: int):
    """Node offsets of a batch's graphs as the host array ``dc_hop_chain_f32`` takes (ascending, covering [0, N]), or
    None.  Unlike ``_segment_arrays`` no per-graph cap applies here - the chain entry has its own (4,096 nodes)."""
    if num_nodes <= 0:
        return None
    nodes = tuple(int(v) for v in segments[0])
    nseg = len(nodes) - 1
    if nseg < 1 or nodes[0] != 0 or nodes[-1] != num_nodes or any(nodes[i + 1] < nodes[i] for i in range(nseg)):
        return None
    import ctypes
    return ((ctypes.c_int64 * (nseg + 1))(*nodes), nseg)


#: DC_VALIDATE=1 graphs built under capture, waiting for their first replay (``validate_pending``)
_PENDING_VALIDATION: list = []
_SHARED_STATUS: dict = {}


def _shared_status(device) -> Optional[torch.Tensor]:
    """The device's shared status word for builds nobody validates (``GraphIndex.__init__``); allocated once, outside any
    hipGraph capture (None while capturing before it exists: the caller then takes a private word, as before)."""
    key = (device.type, device.index)
    t = _SHARED_STATUS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = _SHARED_STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def validate_pending() -> None:
    """DC_VALIDATE=1: check (one device sync) the status words of the adjacencies that were built inside a
    hipGraph capture - call it after the graph's first replay (``train.GraphedTrainStep`` does)."""
    while _PENDING_VALIDATION:
        _PENDING_VALIDATION.pop().validate()


class GraphWindow:
    """Part ``g`` of a merged ``GraphIndex`` seen as a graph of its own: ``fwd`` / ``bwd`` share the
    merged arrays (``ptr`` sliced to the part's rows, ``row_offset`` = the part's first row), so hops
    over the part's OWN feature matrix need no second adjacency.  Part 0 (offset 0) is a valid
    stand-alone adjacency in every respect - its rows, ids and edge order equal those of its own
    ``GraphIndex`` - which is what the fused loss (``ops.contact_losses``) relies on for the soft graph."""

    def __init__(self, merged: GraphIndex, g: int):
        ei, n = merged.parts[g]
        self.merged, self.part = merged, g
        self.edge_index, self.num_nodes, self.num_input_edges = ei, n, int(ei.size(1))
        self.self_loops, self.normalize, self.device = merged.self_loops, merged.normalize, merged.device
        r0 = merged.row_beg[g]
        self.row_offset = r0

        def side(a):
            return SortedAdjacency(a.ptr[r0:r0 + n + 1], a.other, a.perm, a.w, row_offset=r0)
        self.fwd, self.bwd = side(merged.fwd), side(merged.bwd)
        self._static_ok = False
        self._capture_id = merged._capture_id
        self._status = merged._status

    @property
    def _static_ok(self):
        return self.merged._static_ok or self.__dict__.get("_static", False)

    @_static_ok.setter
    def _static_ok(self, v):
        self.__dict__["_static"] = bool(v)
        if v:
            self.merged._static_ok = True

    @property
    def capacity(self) -> int:
        return self.merged.capacity

    def validate(self) -> None:
        self.merged.validate()

    def rebuild(self) -> None:
        self.merged.rebuild()

    def tensors(self):
        out = self.merged.tensors()
        for entry in getattr(self, "_hop_cache", {}).values():
            out += [t for t in entry[:2] if t is not None]
        return out

    def record_stream(self, stream) -> None:
        for t in self.tensors():
            t.record_stream(stream)


# --------------------------------------------------------------------------- #
# cache: one GraphIndex per (edge_index storage, version, shape, N, flags)
# --------------------------------------------------------------------------- #
_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()      # key -> (GraphIndex | GraphWindow, source tensors)
_CACHE_MAX = 32


def _key(edge_index, num_nodes, self_loops, normalize):
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
            tuple(edge_index.stride()), int(num_nodes), bool(self_loops), bool(normalize),
            edge_index.device.index)


def graph_index(edge_index: torch.Tensor, num_nodes: int, *, self_loops: bool = False,
                normalize: bool = True, segments=None) -> GraphIndex:
    """The sorted adjacency of ``edge_index``, built on first use and cached on the tensor's
    address + version + shape.

    The key cannot see writes that bypass PyTorch's version counter (raw kernels, hipGraph replay
    into a static buffer), so an entry is only reused by the stream capture that built it (or, for
    eagerly built entries, outside any capture): a captured step always contains the build of the
    ``edge_index`` buffers it reads, and replays it on whatever those buffers then hold.  Callers
    that guarantee a constant topology for the life of a captured graph may set
    ``GraphIndex._static_ok`` on an eagerly built entry to keep the build out of the capture.

    ``segments`` = ``Batch.segments()`` of the batch this ``edge_index`` belongs to (host-side node / edge offsets
    of its graphs) selects the one-launch ``dc_graph_build_segmented`` when the layout qualifies; the arrays are the
    same bit for bit.
    """
    key = _key(edge_index, num_nodes, self_loops, normalize)
    g = _cache_get(key, edge_index.device if edge_index.is_cuda else None)
    if g is not None:
        return g
    if segments is None:
        segments = edge_layout(edge_index)       # the layout travels on the tensor (data.Batch._tag_edge_layout)
    g = GraphIndex(edge_index, num_nodes, self_loops=self_loops, normalize=normalize, segments=segments)
    _cache_put(key, g, (edge_index,))
    return g


def edge_layout(edge_index: torch.Tensor):
    """The batch layout ``data.Batch`` attached to this ``edge_index`` tensor (``Batch.from_data_list`` / ``.to()`` /
    ``.clone()`` / ``assume_segments``), or None: no tag, or the tensor has been written in place since it was tagged
    (version counter; a tag set by ``assume_segments`` holds for whatever the buffer receives - the caller's promise)."""
    tag = getattr(edge_index, "_dc_segments", None)
    if tag is None:
        return None
    seg, version = tag
    if version is not None and version != edge_index._version:
        return None
    return seg


def _cache_get(key, device):
    hit = _CACHE.get(key)
    if hit is None:
        return None
    g = hit[0]
    cid = capture_id(device) if device is not None else 0
    if g._capture_id == cid or (cid != 0 and g._static_ok):
        _CACHE.move_to_end(key)
        return g
    return None


def _cache_put(key, g, sources) -> None:
    # the entry holds the tensors whose addresses are part of its key: an address cannot be recycled
    # (and matched by a different tensor) while the key is cached - also for aliases added by register()
    _CACHE[key] = (g, tuple(sources))
    _CACHE.move_to_end(key)
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)


def merged_graph_index(parts, *, self_loops: bool = False, normalize: bool = True) -> "GraphIndex":
    """``GraphIndex.from_parts(parts)``, cached like ``graph_index`` (on every part's address + version +
    shape and the capture that built it).  Also makes ``window(g)`` the cached adjacency of part ``g``'s
    ``edge_index``, so per-part code (``conv(x, edge_index)`` of a first layer, the loss) finds it
    instead of building a second adjacency."""
    key = ("merged",) + tuple(_key(ei, n, self_loops, normalize) for ei, n in parts)
    dev = parts[0][0].device if parts[0][0].is_cuda else None
    g = _cache_get(key, dev)
    if g is not None:
        return g
    g = GraphIndex.from_parts(parts, self_loops=self_loops, normalize=normalize)
    _cache_put(key, g, [ei for ei, _ in parts])
    for i, (ei, n) in enumerate(parts):
        w = g.window(i)
        _cache_put(_key(ei, n, self_loops, normalize), w, (ei,))
    return g


def content_hash(t: torch.Tensor) -> int:
    """64-bit content hash of an int64 device tensor (``dc_hash_i64``; synchronises on the
    current stream): the key of ``loaders.TopologyCache``."""
    _require_cuda(t, "tensor")
    if t.dtype != torch.int64:
        raise ValueError("content_hash: int64 tensor expected")
    t = t.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=t.device)
    rc = _lib.lib().dc_hash_i64(t.data_ptr(), t.numel(), out.data_ptr(), current_stream_ptr(t.device))
    _lib.check(rc, "dc_hash_i64")
    return int(out.item()) & 0xFFFFFFFFFFFFFFFF


class NodeOrder:
    """A relabelling of the nodes of one graph: ``perm[new] = old``, ``inv[old] = new`` (int32, on the device).

    Message passing commutes with it: run the convs on ``apply(x)`` / ``relabel(edge_index)`` and
    ``undo`` the result.  The sorted adjacency orders every node's neighbours by EDGE id, which the
    relabelling leaves alone, so the per-row sums - and therefore the outputs - are bit-identical
    to the unordered computation; only where rows live in memory changes.  ``morton(pos)`` sorts
    by Z-order code: for a radius graph over an unordered point cloud
    (``/root/reference/utils/pointcloud_utils.py:7-13``) that turns the hop's neighbour gathers from
    random reads of the whole feature matrix into reads of nearby rows (one XCD's L2).
    Everything runs on the library's kernels (``dc_morton_order`` / ``dc_relabel_edges`` / ``dc_gather_rows``),
    without a host synchronisation: ordering a new cloud can sit inside a captured hipGraph."""

    def __init__(self, perm: torch.Tensor, inv: Optional[torch.Tensor] = None):
        self.perm = perm.to(torch.int32).contiguous()
        if inv is None:
            # a caller's own permutation (``morton`` passes the inverse its kernel built): checked ONCE here - the
            # kernels index with it unchecked (index_select / indexed assignment used to raise on a bad one)
            n = self.perm.numel()
            p64 = self.perm.long()
            # (the check reads three scalars back on the host: under hipGraph capture - where a read-back would invalidate
            # the capture - a caller's permutation is taken on trust, as `morton`'s own always is)
            capturing = self.perm.is_cuda and torch.cuda.is_current_stream_capturing()
            if n and not capturing and (int(p64.min()) < 0 or int(p64.max()) >= n
                                        or int(torch.bincount(p64, minlength=n).max()) != 1):
                raise IndexError(f"NodeOrder: perm is not a permutation of 0..{n - 1}")
            inv = torch.empty_like(self.perm)
            inv[p64] = torch.arange(n, device=self.perm.device, dtype=torch.int32)
        self.inv = inv.to(torch.int32).contiguous()

    @classmethod
    def morton(cls, pos: torch.Tensor) -> "NodeOrder":
        _require_cuda(pos, "pos")
        if pos.dim() != 2 or pos.size(1) < 3 or pos.dtype != torch.float32:
            raise ValueError("NodeOrder.morton: pos must be float32 [N, >=3]")
        n = pos.size(0)
        pos = pos if pos.stride(1) == 1 else pos.contiguous()
        L = _lib.lib()
        perm = torch.empty(n, dtype=torch.int32, device=pos.device)
        inv = torch.empty(n, dtype=torch.int32, device=pos.device)
        ws = torch.empty(max(int(L.dc_morton_order_workspace_bytes(n)), 16), dtype=torch.uint8, device=pos.device)
        _lib.check(L.dc_morton_order(pos.data_ptr(), pos.stride(0), n, perm.data_ptr(), inv.data_ptr(), ws.data_ptr(),
                                     ws.numel(), current_stream_ptr(pos.device)), "dc_morton_order")
        return cls(perm, inv)

    def relabel(self, edge_index: torch.Tensor) -> torch.Tensor:
        _require_cuda(edge_index, "edge_index")
        ei = edge_index.contiguous()
        out = torch.empty_like(ei)
        _lib.check(_lib.lib().dc_relabel_edges(ei.data_ptr(), ei.numel(), self.inv.data_ptr(), self.inv.numel(),
                                               out.data_ptr(), current_stream_ptr(ei.device)), "dc_relabel_edges")
        return out

    @staticmethod
    def _gather_raw(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        _require_cuda(x, "x")
        if x.dim() != 2 or x.size(0) != idx.numel():
            raise ValueError("NodeOrder: x must be [N, F] with N = number of nodes")
        x = x if x.stride(1) == 1 else x.contiguous()
        esz = x.element_size()
        row_bytes, ld_bytes = x.size(1) * esz, x.stride(0) * esz
        if row_bytes % 16 or ld_bytes % 16 or x.data_ptr() % 16:
            return x.index_select(0, idx.long())              # odd widths: stock gather
        out = torch.empty((x.size(0), x.size(1)), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().dc_gather_rows(x.data_ptr(), ld_bytes, idx.data_ptr(), out.data_ptr(), row_bytes,
                                             x.size(0), row_bytes, current_stream_ptr(x.device)), "dc_gather_rows")
        return out

    def _gather(self, x: torch.Tensor, idx: torch.Tensor, idx_back: torch.Tensor) -> torch.Tensor:
        """``x[idx]``; differentiable: ``idx`` and ``idx_back`` are inverse permutations, so the gradient of a row
        gather is the row gather of the gradient by the opposite permutation (``index_select``'s backward,
        without its ``index_add_`` atomics)."""
        from .deferred import resolve
        x = resolve(x)                     # (a plain conv call's deferred result: Function.apply does not dispatch)
        if torch.is_grad_enabled() and x.requires_grad:
            return _PermuteRows.apply(x, idx, idx_back)
        return self._gather_raw(x, idx)

    def apply(self, x: torch.Tensor) -> torch.Tensor:
        return self._gather(x, self.perm, self.inv)

    def undo(self, y: torch.Tensor) -> torch.Tensor:
        return self._gather(y, self.inv, self.perm)


class _PermuteRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, idx_back):
        ctx.idx_back = idx_back
        return NodeOrder._gather_raw(x, idx)

    @staticmethod
    def backward(ctx, gy):
        return NodeOrder._gather_raw(gy.contiguous(), ctx.idx_back), None, None


def register(edge_index: torch.Tensor, g: GraphIndex) -> None:
    """Make ``g`` (built elsewhere, e.g. by ``loaders.TopologyCache`` on the loader's stream) the
    cached adjacency of ``edge_index``.  Every alias keeps its own ``edge_index`` alive for as long as
    its key is cached (one ``g`` may be registered for many tensors)."""
    g._capture_id = 0
    _cache_put(_key(edge_index, g.num_nodes, g.self_loops, g.normalize), g, (edge_index,))


def clear_cache() -> None:
    _CACHE.clear()
    _PENDING_VALIDATION.clear()          # (DC_VALIDATE=1 graphs of captures that were never replayed)
