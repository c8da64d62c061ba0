"""``Data`` / ``Batch``: the slice of ``torch_geometric.data`` the reference touches.

Used by the reference at ``utils/graph_utils.py:20`` (``Data(x=, edge_index=, pos=)``),
``train.py:36-44`` / ``eval.py:53-54,107-109`` (``Batch.from_data_list(...).to(device)``),
``models/model.py:69,71,75,77,91-95`` (``.x``, ``.edge_index``, ``.clone()``, ``.pos +=``)
and ``eval.py:149,158`` (``batch[i]``).
"""
from .batch import Batch, Data  # noqa: F401

__all__ = ["Data", "Batch"]
