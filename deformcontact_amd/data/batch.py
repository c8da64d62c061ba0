from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch
from torch import Tensor


class Data:
    """Attribute bag for one graph: ``x [N,F]``, ``edge_index [2,E]`` int64, ``pos [N,3]``
    plus arbitrary extra attributes (tensors move with ``to()`` / copy with ``clone()``)."""

    def __init__(self, x: Optional[Tensor] = None, edge_index: Optional[Tensor] = None,
                 pos: Optional[Tensor] = None, **kwargs: Any):
        self.x, self.edge_index, self.pos = x, edge_index, pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- introspection ------------------------------------------------------
    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    @property
    def num_nodes(self) -> int:
        for t in (self.x, self.pos):
            if isinstance(t, Tensor):
                return t.size(0)
        if isinstance(self.edge_index, Tensor) and self.edge_index.numel() > 0:
            return int(self.edge_index.max()) + 1
        return 0

    @property
    def num_edges(self) -> int:
        return 0 if self.edge_index is None else int(self.edge_index.size(1))

    # -- copies / moves -----------------------------------------------------
    def _apply(self, fn):
        for k, v in self.__dict__.items():
            if isinstance(v, Tensor):
                self.__dict__[k] = fn(v)
        self._tag_edge_layout()
        return self

    def _tag_edge_layout(self, sticky: bool = False) -> None:
        """Attach the batch layout (``Batch.segments()``) to the ``edge_index`` TENSOR: the reference hands the conv only
        ``graph.edge_index`` (``models/model.py:71,77``: ``conv(x, graph.edge_index)``), so that is where
        ``graph.graph_index`` looks for it (``graph.edge_layout``) when no ``segments`` are passed - the one-launch
        segmented adjacency build and ``dc_hop_chain_f32`` are then reached by the unchanged reference wiring.  The tag
        carries the tensor's version counter and is ignored once the tensor has been written in place (``sticky``:
        the caller vouches for every later content, ``Batch.assume_segments``)."""
        seg, ei = self.__dict__.get("_segments"), self.__dict__.get("edge_index")
        if seg is not None and isinstance(ei, Tensor):
            ei._dc_segments = (seg, None if sticky else ei._version)

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        out.__dict__ = {k: (v.clone() if isinstance(v, Tensor) else
                            (list(v) if isinstance(v, list) else v))
                        for k, v in self.__dict__.items()}
        if isinstance(self.edge_index, Tensor):
            # provenance of the copy's edge set (tensor it was cloned from + its version): lets train.losses
            # know, without reading device memory, that a prediction shares the rest batch's edges
            out.__dict__["_dc_cloned_edges"] = (self.edge_index, self.edge_index._version)
        out._tag_edge_layout()
        return out

    def to(self, device, non_blocking: bool = False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def __repr__(self) -> str:
        parts = [f"{k}={list(v.shape)}" if isinstance(v, Tensor) else f"{k}={v!r}"
                 for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Block-diagonal union of graphs (``Batch.from_data_list``).

    Node-level tensors (first dim == ``num_nodes``) are concatenated on dim 0,
    ``edge_index`` on dim 1 with the cumulative node offset added, other tensors
    are stacked on a new dim 0, non-tensors collected into lists.  ``batch [N]``
    maps nodes to graphs, ``ptr [B+1]`` are node offsets.  The concatenation
    order IS the edge order the sorted adjacency reproduces bit-exactly.
    """

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data]) -> "Batch":
        if len(data_list) == 0:
            raise ValueError("Batch.from_data_list: empty list")
        out = cls()
        counts = [d.num_nodes for d in data_list]
        ecounts = [d.num_edges for d in data_list]
        dev = None
        for d in data_list:
            for v in d.__dict__.values():
                if isinstance(v, Tensor):
                    dev = v.device
                    break
            if dev is not None:
                break
        ptr = torch.zeros(len(counts) + 1, dtype=torch.long)
        ptr[1:] = torch.tensor(counts, dtype=torch.long).cumsum(0)
        eptr = torch.zeros(len(counts) + 1, dtype=torch.long)
        eptr[1:] = torch.tensor(ecounts, dtype=torch.long).cumsum(0)
        offs = ptr.tolist()

        keys = [k for k in data_list[0].__dict__ if not k.startswith("_")]
        kinds: Dict[str, str] = {}
        for k in keys:
            vals = [getattr(d, k, None) for d in data_list]
            v0 = vals[0]
            if v0 is None:
                setattr(out, k, None)
                continue
            if k == "edge_index":
                setattr(out, k, torch.cat([v + offs[i] for i, v in enumerate(vals)], dim=1))
                kinds[k] = "edge"
            elif isinstance(v0, Tensor) and v0.dim() >= 1 and all(
                    isinstance(v, Tensor) and v.size(0) == c for v, c in zip(vals, counts)):
                setattr(out, k, torch.cat(vals, dim=0))
                kinds[k] = "node"
            elif isinstance(v0, Tensor):
                setattr(out, k, torch.stack(vals, dim=0))
                kinds[k] = "graph"
            else:
                setattr(out, k, list(vals))
                kinds[k] = "list"
        out.batch = torch.repeat_interleave(
            torch.arange(len(counts)), torch.tensor(counts, dtype=torch.long)).to(dev)
        out.ptr = ptr.to(dev)
        out._edge_ptr = eptr
        out._node_ptr = ptr.clone()
        out._kinds = kinds
        # host-side layout (node / edge offsets of the graphs): survives to() / clone() as plain tuples
        out.__dict__["_segments"] = (tuple(offs), tuple(eptr.tolist()))
        out._tag_edge_layout()
        return out

    def __setattr__(self, name, value):
        # a replaced edge_index (radius graph, re-meshing) no longer follows the recorded layout
        # - and what was known about its equality with another batch's edges (train.losses picks the fused loss by it)
        if name == "edge_index":
            for k in ("_segments", "_dc_edges_equal", "_dc_cloned_edges"):
                self.__dict__.pop(k, None)
        object.__setattr__(self, name, value)

    def assume_segments(self, segments) -> None:
        """Declare that the CURRENT ``edge_index`` (e.g. a view into a static buffer that receives batches of one
        fixed layout) follows ``segments`` = an earlier ``segments()`` value."""
        self.__dict__["_segments"] = (tuple(int(v) for v in segments[0]), tuple(int(v) for v in segments[1]))
        self._tag_edge_layout(sticky=True)

    def segments(self):
        """``(node offsets, edge offsets)`` of the graphs in this batch as host tuples - graph ``i`` owns nodes
        ``[n[i], n[i+1])`` and edges ``[e[i], e[i+1])`` of ``edge_index`` - or None once ``edge_index`` has been
        replaced.  ``graph.graph_index(..., segments=)`` uses it to build the sorted adjacency in one launch."""
        return self.__dict__.get("_segments")

    @property
    def num_graphs(self) -> int:
        return int(self._node_ptr.numel()) - 1

    def __len__(self) -> int:
        return self.num_graphs

    def get_example(self, i: int) -> Data:
        b = self.num_graphs
        if i < 0:
            i += b
        if not 0 <= i < b:
            raise IndexError(f"graph index {i} out of range for a batch of {b}")
        a, z = int(self._node_ptr[i]), int(self._node_ptr[i + 1])
        ea, ez = int(self._edge_ptr[i]), int(self._edge_ptr[i + 1])
        d = Data()
        for k, kind in self._kinds.items():
            v = getattr(self, k)
            if kind == "edge":
                d.edge_index = v[:, ea:ez] - a
            elif kind == "node":
                setattr(d, k, v[a:z])
            else:
                setattr(d, k, v[i])
        return d

    def __getitem__(self, i):
        if isinstance(i, int):
            return self.get_example(i)
        raise TypeError("Batch supports integer indexing only")

    def to_data_list(self) -> List[Data]:
        return [self.get_example(i) for i in range(self.num_graphs)]
