"""PyG-shaped conv operators backed by the gfx950 C-ABI library.

Same class names, constructor signature ``Conv(in_channels, out_channels)``, call
``conv(x, edge_index)`` and ``state_dict`` keys as ``torch_geometric.nn`` 2.5.2,
which is what ``/root/reference/models/model.py:2,39,45,49,71,77`` uses.
"""
from .conv import GATConv, GCNConv, TAGConv, knn  # noqa: F401

__all__ = ["TAGConv", "GCNConv", "GATConv", "knn"]
